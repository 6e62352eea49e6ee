// tu_team.hip -- K2, the fused wavefront sweeps with a helper wavefront per item (riccati_mfma.hpp, HELP): launches of at most
// one item per SIMD -- a job's draining tail, a single small batch -- where a lone wavefront issues at half rate.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "launch.hpp"
#include "riccati_mfma.hpp"

namespace dpilqr {

#define DPILQR_TEAM_SIZES(X) X(4, 2) X(8, 4) X(12, 6) X(16, 8) X(20, 10)

// Returns DPILQR_EUNSUPPORTED (without touching the error text) when the batch has no team instantiation; the caller then
// launches the one-wavefront-per-item kernel.
int32_t launch_riccati_team(const dpilqr_batch_desc& D, const double* X, const double* U, const double* mu, double* K, double* d,
                            int32_t* singular, const int32_t* items, const int32_t* n_items, int grid_items, int gains_by_item,
                            hipStream_t st) {
    static const bool off = route_flag("DPILQR_NO_TEAM");   // A/B switch
    if (off || grid_items <= 0) return DPILQR_EUNSUPPORTED;
    const int n = D.k * D.n_s, m = D.k * D.n_c;
    const int cus = device_cus();
    const int grid = grid_items <= cus ? grid_items : (grid_items + cus * 4 - 1) / (cus * 4) * cus;
#define DPILQR_TRY_TEAM(NN, MM, FU)                                                                                \
    if (n == NN && m == MM) {                                                                                      \
        using CF = MfmaCfg<NN, MM, FU, true>;                                                                      \
        const size_t lds_t = sizeof(double) * CF::total * 4;                                                       \
        auto kern = k_riccati_mfma_team<NN, MM, FU>;                                                               \
        int32_t rc_t = allow_lds(kern, lds_t);                                                                     \
        if (rc_t) return rc_t;                                                                                     \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_t, st, D.B, D.T, mu, K, d, singular, items, n_items,    \
                           gains_by_item, cus, FusedArgs{D, X, U});                                                \
        HIP_TRY(hipGetLastError());                                                                                \
        return DPILQR_OK;                                                                                          \
    }
#define DPILQR_TRY_TEAM1(NN, MM) DPILQR_TRY_TEAM(NN, MM, 1)
#define DPILQR_TRY_TEAM2(NN, MM) DPILQR_TRY_TEAM(NN, MM, 2)
    if (fused_wavefront_sweep_applies(D)) {
        DPILQR_TEAM_SIZES(DPILQR_TRY_TEAM1)
    } else if (fused_wavefront_general_applies(D)) {
        DPILQR_TEAM_SIZES(DPILQR_TRY_TEAM2)
    }
    return DPILQR_EUNSUPPORTED;
}

}  // namespace dpilqr
