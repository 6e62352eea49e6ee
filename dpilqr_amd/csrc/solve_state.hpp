// solve_state.hpp -- the per-item solver state the line-search kernels update (control.py:150-225); plain structs
// shared by the host loop (dpilqr_hip.hip) and the kernels of tu_forward.hip / tu_big.hip.
#pragma once
#include <cstdint>

#include "dpilqr_hip.h"

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
    const int32_t* singular;  // [B] or null: 1 = a zero pivot, 2 = the sweep gave the item up (riccati_big.hpp: a team hand-over expired)
    int32_t* fault;           // one word or null: items retired with DPILQR_STATUS_FAULT so far (dpilqr_solve_batch reports them)
    int32_t* next_count;      // number of items pushed onto next_items so far
    int32_t* next_items;      // active list of the next iteration
    int32_t n_lqr_iter;
    int32_t gains_by_item;    // K, d indexed by item id (caller asked for them) instead of list position
    double tol;
    // t_kill (control.py:213-218): per-item admission stamps of the constant-rate clock and the limit in its ticks (0: none)
    const int64_t* t_admit;
    int64_t t_kill_ticks;
};

// The reference's real-time bail-out, decided where the accept / converge decision is: called for an item whose step was
// accepted and did not converge.  perf_counter() - t0 > t_kill with t0 = the item's admission (just before its first
// backward pass, as control.py:167 takes it after the initial rollout).
__device__ inline bool solve_time_is_up(const SolveState& S, int b) {
    if (S.t_kill_ticks <= 0) return false;
    return (int64_t)__builtin_amdgcn_s_memrealtime() - S.t_admit[b] > S.t_kill_ticks;
}

// An item whose backward pass did not produce gains is retired by the line-search kernel that meets it: np.linalg.solve would
// have raised LinAlgError (singular[b] == 1), or the sweep itself gave the item up (== 2).  Called by ONE thread of the item.
__device__ inline void retire_without_gains(const SolveState& S, int b) {
    const bool fault = S.singular[b] == 2;
    S.status[b] = fault ? DPILQR_STATUS_FAULT : DPILQR_STATUS_SINGULAR;
    S.n_bwd[b] += 1;
    if (fault && S.fault) atomicAdd(S.fault, 1);
}

}  // namespace dpilqr
