// tu_lsteam.hip -- K3 for launches of at most one item per SIMD: the line search with a team of two wavefronts per item
// (forward_team.hpp).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "launch.hpp"
#include "forward_team.hpp"

namespace dpilqr {

// Returns DPILQR_EUNSUPPORTED (without touching the error text) when the batch has no instantiation or the launch is too large
// to gain; the caller then launches the one-wavefront kernel.
int32_t launch_linesearch_team(const dpilqr_batch_desc& D, double* X, double* U, const double* K, const double* d,
                               const double* alphas, double* Xc, double* Uc, const SolveState& S, const int32_t* items,
                               const int32_t* n_items, int grid_items, hipStream_t st) {
    static const bool off = route_flag("DPILQR_LS_NO_TEAM");   // A/B switch
    static const int max_items = route_int("DPILQR_LS_TEAM_MAX", 1024);
    if (off || grid_items > max_items || grid_items <= 0 || hint_model(D) < 0) return DPILQR_EUNSUPPORTED;
    const int model = hint_model(D);
#define DPILQR_TRY_LSTEAM(MODEL, KA)                                                                                \
    if (model == MODEL && D.k == KA && model_ns(MODEL) == D.n_s && model_nc(MODEL) == D.n_c) {                      \
        using TF = TeamFwdLds<MODEL, KA>;                                                                           \
        const size_t lds_t = sizeof(double) * TF::total;                                                            \
        hipLaunchKernelGGL((k_linesearch_team<MODEL, KA>), dim3(grid_items), dim3(128), lds_t, st, D, X, U, K, d,   \
                           alphas, Xc, Uc, S, items, n_items);                                                      \
        HIP_TRY(hipGetLastError());                                                                                 \
        return DPILQR_OK;                                                                                           \
    }
#define DPILQR_LSTEAM_6(MODEL) DPILQR_TRY_LSTEAM(MODEL, 1) DPILQR_TRY_LSTEAM(MODEL, 2) DPILQR_TRY_LSTEAM(MODEL, 3) \
        DPILQR_TRY_LSTEAM(MODEL, 4) DPILQR_TRY_LSTEAM(MODEL, 5) DPILQR_TRY_LSTEAM(MODEL, 6)
    DPILQR_LSTEAM_6(kDoubleInt4D)
    DPILQR_LSTEAM_6(kUnicycle4D)
    DPILQR_LSTEAM_6(kQuadcopter6D)
    DPILQR_LSTEAM_6(kDoubleInt6D)
    DPILQR_LSTEAM_6(kCar3D)
#undef DPILQR_LSTEAM_6
#undef DPILQR_TRY_LSTEAM
    return DPILQR_EUNSUPPORTED;
}

}  // namespace dpilqr
