// tu_bigfwd.hip -- the forward pass / line search of the large-cluster path (n_x > 60: BASELINE config 5) and of the fp32 arm:
// k_forward reading K[t] from global memory instead of staging it in LDS (forward.hpp, KDIRECT), for double and float.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "launch.hpp"
#include "forward.hpp"

namespace dpilqr {

template <typename R>
static int32_t launch_forward_big_t(const dpilqr_batch_desc& D, int mode, const R* x0, R* X, R* U, const R* K, const R* d,
                                    const double* alphas, int ngrp, R* Xc, R* Uc, double* Jc, const SolveState& S,
                                    const int32_t* items, const int32_t* n_items, int grid_items, hipStream_t st) {
    if (grid_items <= 0) return DPILQR_OK;
    const int n = D.k * D.n_s, m = D.k * D.n_c;
    const int groups = mode == kModeRollout ? 1 : ngrp;
    const int threads = ((D.k * groups + 63) / 64) * 64;
    if (threads > 256)
        return fail(DPILQR_EUNSUPPORTED, "k*n_alpha=%d exceeds the 256-thread workgroup of the forward pass", D.k * groups);
    const size_t lds = (forward_lds_bytes(n, m, D.k, ngrp, true, sizeof(R)) + 15) & ~(size_t)15;
    // K[t] dx on the matrix pipe where every wavefront gets at most two row tiles (forward.hpp: forward_on_pipe) -- config 5's
    // shape and every cluster of the reference's families at ten candidates; a rollout has no gains
    const bool pipe = mode != kModeRollout && forward_on_pipe(n, m, D.k, threads, ngrp);
    DISPATCH_FAMILY(D.n_s, {
        if (pipe) {
            int32_t rc = allow_lds(k_forward<R, NS, NC, true, true>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL((k_forward<R, NS, NC, true, true>), dim3(grid_items), dim3(threads), lds, st, D, mode, x0, X, U, K, d,
                               alphas, ngrp, Xc, Uc, Jc, S, items, n_items, 1, (int)(lds / sizeof(R)));
        } else {
            int32_t rc = allow_lds(k_forward<R, NS, NC, true, false>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL((k_forward<R, NS, NC, true, false>), dim3(grid_items), dim3(threads), lds, st, D, mode, x0, X, U, K, d,
                               alphas, ngrp, Xc, Uc, Jc, S, items, n_items, 1, (int)(lds / sizeof(R)));
        }
    })
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t launch_forward_big_f64(const dpilqr_batch_desc& D, int mode, const double* x0, double* X, double* U, const double* K,
                               const double* d, const double* alphas, int ngrp, double* Xc, double* Uc, double* Jc,
                               const SolveState& S, const int32_t* items, const int32_t* n_items, int grid_items,
                               hipStream_t st) {
    return launch_forward_big_t<double>(D, mode, x0, X, U, K, d, alphas, ngrp, Xc, Uc, Jc, S, items, n_items, grid_items, st);
}
int32_t launch_forward_big_f32(const dpilqr_batch_desc& D, int mode, const float* x0, float* X, float* U, const float* K,
                               const float* d, const double* alphas, int ngrp, float* Xc, float* Uc, double* Jc,
                               const SolveState& S, const int32_t* items, const int32_t* n_items, int grid_items,
                               hipStream_t st) {
    return launch_forward_big_t<float>(D, mode, x0, X, U, K, d, alphas, ngrp, Xc, Uc, Jc, S, items, n_items, grid_items, st);
}

}  // namespace dpilqr
