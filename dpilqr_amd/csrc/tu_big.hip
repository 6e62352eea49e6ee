// tu_big.hip -- the path for large clusters (n_x > 60: BASELINE config 5) and the fp32 arm of its tolerance study:
// the fused workgroup-per-item Riccati sweep (riccati_big.hpp) and the forward pass / line search that reads K[t] from
// global memory instead of staging it in LDS (forward.hpp, KDIRECT), both for double and float.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "launch.hpp"
#include "forward.hpp"
#include "riccati_big.hpp"

namespace dpilqr {

int64_t riccati_big_scratch_elems(int n, int m) { return BigScratch(n, m).total; }

template <typename R>
static int32_t launch_riccati_big_t(const dpilqr_batch_desc& D, const R* X, const R* U, const double* mu, R* K, R* d,
                                    int32_t* singular, const int32_t* items, const int32_t* n_items, int grid_items,
                                    int gains_by_item, void* scratch, hipStream_t st) {
    if (grid_items <= 0) return DPILQR_OK;
    const size_t lds = sizeof(R) * (size_t)BigLds(D.k, D.n_s, D.n_c).total;
    DISPATCH_FAMILY(D.n_s, {
        int32_t rc = allow_lds(k_riccati_big<R, NS, NC>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((k_riccati_big<R, NS, NC>), dim3(grid_items), dim3(kBigThreads), lds, st, D, X, U, mu, K, d,
                           singular, items, n_items, gains_by_item, static_cast<R*>(scratch));
    })
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t launch_riccati_big_f64(const dpilqr_batch_desc& D, const double* X, const double* U, const double* mu, double* K,
                               double* d, int32_t* singular, const int32_t* items, const int32_t* n_items,
                               int grid_items, int gains_by_item, void* scratch, hipStream_t st) {
    return launch_riccati_big_t<double>(D, X, U, mu, K, d, singular, items, n_items, grid_items, gains_by_item, scratch, st);
}
int32_t launch_riccati_big_f32(const dpilqr_batch_desc& D, const float* X, const float* U, const double* mu, float* K,
                               float* d, int32_t* singular, const int32_t* items, const int32_t* n_items,
                               int grid_items, int gains_by_item, void* scratch, hipStream_t st) {
    return launch_riccati_big_t<float>(D, X, U, mu, K, d, singular, items, n_items, grid_items, gains_by_item, scratch, st);
}

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
    DISPATCH_FAMILY(D.n_s, {
        int32_t rc = allow_lds(k_forward<R, NS, NC, true>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((k_forward<R, NS, NC, true>), dim3(grid_items), dim3(threads), lds, st, D, mode, x0, X, U, K, d,
                           alphas, ngrp, Xc, Uc, Jc, S, items, n_items, 1, (int)(lds / sizeof(R)));
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
