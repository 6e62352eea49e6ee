// tu_inprod.hip -- K2, the wavefront sweeps with in-sweep production (riccati_mfma.hpp, PNS): the record-free form for clusters
// of at most four agents of the six-state family, at most six CarDynamics3D agents (padded into the next instantiated size) and
// the four-state clusters the fused forms of tu_riccati.hip do not serve.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "launch.hpp"
#include "riccati_mfma.hpp"

namespace dpilqr {

// Returns DPILQR_EUNSUPPORTED (without touching the error text) when the batch has no instantiation.
int32_t launch_riccati_inprod(const dpilqr_batch_desc& D, const double* X, const double* U, const double* mu, double* K, double* d,
                              int32_t* singular, const int32_t* items, const int32_t* n_items, int grid_items, int gains_by_item,
                              hipStream_t st) {
    if (grid_items <= 0 || !fused_wavefront_inprod_applies(D)) return DPILQR_EUNSUPPORTED;
    const int n = D.k * D.n_s, m = D.k * D.n_c;
    static const int max_wv = route_int("DPILQR_MFMA_WAVES", 8);
    const int cus = device_cus();
#define DPILQR_TRY_INPROD(NN, MM, PNS_)                                                                            \
    if (D.n_s == PNS_ && n <= NN && m <= MM) {                                                                     \
        static_assert(MfmaCfg<NN, MM>::supported, "MFMA sweep not available for this size");                       \
        constexpr size_t per_wave = sizeof(double) * (MfmaCfg<NN, MM>::total + InprodCfg<NN, MM, PNS_>::total);    \
        const int wv = (grid_items > 1024 && max_wv >= 8 && per_wave * 8 <= (size_t)kMaxLds) ? 8 : 4;              \
        const size_t lds_t = per_wave * wv;                                                                        \
        constexpr bool PAD_ = (NN % PNS_ != 0) || (MM * PNS_ != NN * InprodCfg<NN, MM, PNS_>::PNC);                \
        /* (the sizes are tried in ascending order: an exact-size instantiation only ever sees clusters of exactly its size) */ \
        auto kern = wv == 8 ? k_riccati_mfma_inprod<NN, MM, 8, PNS_, PAD_> : k_riccati_mfma_inprod<NN, MM, 4, PNS_, PAD_>;  \
        int32_t rc_t = allow_lds(kern, lds_t);                                                                     \
        if (rc_t) return rc_t;                                                                                     \
        const int grid = grid_items <= cus ? grid_items : (grid_items + cus * wv - 1) / (cus * wv) * cus;          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wv), lds_t, st, D.B, D.T, mu, K, d, singular, items,         \
                           n_items, gains_by_item, cus, FusedArgs{D, X, U}, n, m);                                 \
        HIP_TRY(hipGetLastError());                                                                                \
        g_sweep_waves = wv;                                                                                        \
        return DPILQR_OK;                                                                                          \
    }
    // six-state family: one agent (6, 3) -> (8, 4); two (12, 6); three (18, 9) -> (20, 10); four (24, 12)
    DPILQR_TRY_INPROD(8, 4, 6) DPILQR_TRY_INPROD(12, 6, 6) DPILQR_TRY_INPROD(20, 10, 6) DPILQR_TRY_INPROD(24, 12, 6)
    // CarDynamics3D: one agent (3, 2) -> (4, 2); two (6, 4) -> (8, 4); three (9, 6) -> (12, 6); four (12, 8) -> (16, 8);
    // five (15, 10) -> (20, 10); six (18, 12) -> (24, 12)
    DPILQR_TRY_INPROD(4, 2, 3) DPILQR_TRY_INPROD(8, 4, 3) DPILQR_TRY_INPROD(12, 6, 3) DPILQR_TRY_INPROD(16, 8, 3)
    DPILQR_TRY_INPROD(20, 10, 3) DPILQR_TRY_INPROD(24, 12, 3)
    // four-state family (exact sizes): clusters of at most five agents without the fused forms' hints
    DPILQR_TRY_INPROD(4, 2, 4) DPILQR_TRY_INPROD(8, 4, 4) DPILQR_TRY_INPROD(12, 6, 4) DPILQR_TRY_INPROD(16, 8, 4)
    DPILQR_TRY_INPROD(20, 10, 4)
    // (One twelve-state agent, (12, 4) -> (12, 6), was built and measured: 255 registers, bit-identical -- and slower: 1.00 ms per
    // 512 items and pass against 0.27 + 0.43, sweeps of a four-iteration solve 4.1 against 2.1 ms: Quadcopter12D's Jacobian is
    // most of such a step and serial in t here, parallel over t in the producer.  profiles/r04_inprod_twelve_state.txt)
#undef DPILQR_TRY_INPROD
    return DPILQR_EUNSUPPORTED;
}

}  // namespace dpilqr
