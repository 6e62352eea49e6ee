// solve_state.hpp -- the per-item solver state the line-search kernels update (control.py:150-225); plain structs
// shared by the host loop (dpilqr_hip.hip) and the kernels of tu_forward.hip / tu_big.hip.
#pragma once
#include <cstdint>

namespace dpilqr {

enum ForwardMode : int { kModeRollout = 0, kModeCandidates = 1, kModeLineSearch = 2 };

struct SolveState {  // per-item solver state, device arrays of length B
    double* mu;
    double* delta;
    double* J_star;
    double* J_last;
    int32_t* status;
    int32_t* n_bwd;
    int32_t* n_fwd;
    double* trace;            // [B][n_lqr_iter][5] or null
    const int32_t* singular;  // [B] or null
    int32_t* next_count;      // number of items pushed onto next_items so far
    int32_t* next_items;      // active list of the next iteration
    int32_t n_lqr_iter;
    int32_t gains_by_item;    // K, d indexed by item id (caller asked for them) instead of list position
    double tol;
};

}  // namespace dpilqr
