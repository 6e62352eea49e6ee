// tu_big.hip -- the path for large clusters (n_x > 60: BASELINE config 5) and the fp32 arm of its tolerance study:
// the fused workgroup-per-item Riccati sweep (riccati_big.hpp), for double and float.  (Its forward pass / line search is
// tu_bigfwd.hip: this unit is compiled with machine LICM off -- __graft_entry__.UNIT_FLAGS -- which only the sweep needs.)
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "launch.hpp"
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

}  // namespace dpilqr
