// tu_tiles.hip -- K1, the tile producers (tiles.hpp, tiles_wave.hpp), and their launcher.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "launch.hpp"
#include "tiles.hpp"
#include "tiles_wave.hpp"

namespace dpilqr {

int32_t launch_make_tiles(const dpilqr_batch_desc& D, const double* X, const double* U, double* tiles,
                          const int32_t* items, const int32_t* n_items, int grid_items, bool sparse, bool dyn_only,
                          hipStream_t st) {
    if (grid_items <= 0) return DPILQR_OK;
    static const bool force_dense = route_flag("DPILQR_TILES_DENSE");   // A/B switch
    if (force_dense) sparse = false;
    if (!sparse) dyn_only = false;
    // the solve loop's producer for a batch of one linear model whose (X, U)-independent entries are already in place:
    // kernels compiled per (model, agents), tiles_wave.hpp.  (Measured: for the other cases -- A, B to be written too,
    // or more than 6 agents -- the generic producer's sparse stores are the faster ones.)
    static const bool no_wave = route_flag("DPILQR_TILES_GENERIC");   // A/B switch
    if (sparse && dyn_only && !no_wave && hint_model(D) >= 0) {
        const int model = hint_model(D);
        // rows of L_xx beyond the proximity cost's dimensions hold w_ref (Q + Q^T) only: with one Q, Q_f for the batch they
        // were placed with A, B, L_uu and are skipped as well
        const int und = hint_n_dims(D);
        const int xx_rows = (D.Q_bstride == 0 && D.Qf_bstride == 0 && und >= 1 && und < D.n_s) ? und : D.n_s;
#define DPILQR_TRY_TW(MODEL, KA, LINEAR)                                                                            \
    if (model == MODEL && D.k == KA && model_ns(MODEL) == D.n_s && model_nc(MODEL) == D.n_c) {                      \
        constexpr int rpg = TilesWaveCfg<MODEL, KA, false>::RPG;                                                    \
        const int n_groups = (D.T + 1 + rpg - 1) / rpg;                                                             \
        const int gpw = 1;   /* one wavefront per group of records: measured against 2, 3, 5, 9 groups per wavefront */ \
        const dim3 grid_w((n_groups + gpw - 1) / gpw, grid_items);                                                  \
        const size_t lds_w = sizeof(double) * TilesWaveCfg<MODEL, KA, true>::total;                                 \
        hipLaunchKernelGGL((k_make_tiles_wave<MODEL, KA, true>), grid_w, dim3(64), lds_w, st, D, X, U, tiles, items, \
                           n_items, gpw, xx_rows);                                                                  \
        HIP_TRY(hipGetLastError());                                                                                 \
        return DPILQR_OK;                                                                                           \
    }
#define DPILQR_TW_6(MODEL, LINEAR) DPILQR_TRY_TW(MODEL, 1, LINEAR) DPILQR_TRY_TW(MODEL, 2, LINEAR)                  \
        DPILQR_TRY_TW(MODEL, 3, LINEAR) DPILQR_TRY_TW(MODEL, 4, LINEAR) DPILQR_TRY_TW(MODEL, 5, LINEAR)             \
        DPILQR_TRY_TW(MODEL, 6, LINEAR)
        DPILQR_TW_6(kDoubleInt4D, true)
#undef DPILQR_TW_6
#undef DPILQR_TRY_TW
    }
    const int ts = make_tiles_steps(D.k, D.n_s, D.n_c);
    const size_t lds = make_tiles_lds_bytes(D.k, D.n_s, D.n_c, ts);
    dim3 grid((D.T + 1 + ts - 1) / ts, grid_items);
    DISPATCH_FAMILY(D.n_s, {
        int32_t rc = allow_lds(k_make_tiles<NS, NC, false>, lds);
        if (rc) return rc;
        if (sparse) {
            if ((rc = allow_lds(k_make_tiles<NS, NC, true>, lds))) return rc;
            hipLaunchKernelGGL((k_make_tiles<NS, NC, true>), grid, dim3(64), lds, st, D, X, U, tiles, items, n_items, ts,
                               dyn_only ? 1 : 0);
        } else {
            hipLaunchKernelGGL((k_make_tiles<NS, NC, false>), grid, dim3(64), lds, st, D, X, U, tiles, items, n_items, ts, 0);
        }
    })
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t tile_layout_host(int32_t n_x, int32_t n_u, int64_t offsets[7], int64_t row_strides[7], int64_t* stride) {
    const TileLayout L(n_x, n_u);
    offsets[0] = L.oA; offsets[1] = L.oB; offsets[2] = L.oLxx; offsets[3] = L.oLux; offsets[4] = L.oLuu;
    offsets[5] = L.oLx; offsets[6] = L.oLu;
    row_strides[0] = L.ldAB; row_strides[1] = L.ldAB; row_strides[2] = n_x; row_strides[3] = L.ldUG; row_strides[4] = L.ldUG;
    row_strides[5] = 1; row_strides[6] = 1;
    *stride = L.stride;
    return DPILQR_OK;
}

}  // namespace dpilqr
