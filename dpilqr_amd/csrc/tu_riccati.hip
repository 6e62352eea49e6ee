// tu_riccati.hip -- K2, the Riccati sweeps for n_x <= 60 (riccati_mfma.hpp, riccati_wg.hpp, riccati_tiled.hpp,
// riccati.hpp), and their launcher.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "launch.hpp"
#include "riccati.hpp"
#include "riccati_mfma.hpp"
#include "riccati_tiled.hpp"
#include "riccati_wg.hpp"

namespace dpilqr {

static int riccati_threads(int n) { return n <= 24 ? 64 : (n <= 36 ? 128 : 256); }

// compile-time-sized sweeps (one wavefront per sub-problem); everything else takes the generic kernel
#define DPILQR_TILED_SIZES(X) X(4, 2) X(8, 4) X(12, 6) X(16, 8) X(20, 10)

thread_local int g_sweep_waves = 0;

int32_t launch_riccati(int B, int T, int n, int m, const double* tiles, const double* mu, double* K, double* d,
                       int32_t* singular, const int32_t* items, const int32_t* n_items, int grid_items,
                       int gains_by_item, int block_ns, int block_nc, hipStream_t st) {
    g_sweep_waves = 0;
    if (grid_items <= 0) return DPILQR_OK;
    // block_ns > 0: the caller guarantees that [A|B] is block diagonal with block_ns x (block_ns + block_nc) blocks
    // (tiles made by k_make_tiles from a MultiDynamicalModel); 0: arbitrary dense tiles (the plugin boundary).
    static const bool no_bd = route_flag("DPILQR_RICCATI_DENSE");   // A/B switch
    const bool bd = !no_bd && block_ns == 4 && block_nc == 2 && n == 4 * (m / 2) && m % 2 == 0;
    // sweep selection: matrix-pipe kernel where instantiated, else the vector-pipe tiled kernel, else the generic one
    // (DPILQR_RICCATI=mfma|tiled|generic pins one for A/B measurements)
    static const char* pick_env = route_env("DPILQR_RICCATI");
    static const int pick = route_flag("DPILQR_FORCE_GENERIC_RICCATI") ? 2
                            : (!pick_env ? 0 : (!strcmp(pick_env, "tiled") ? 1 : (!strcmp(pick_env, "generic") ? 2 : 0)));
    if (pick == 0) {
#define DPILQR_TRY_MFMA(NN, MM)                                                                                    \
    if (n == NN && m == MM) {                                                                                      \
        static_assert(MfmaCfg<NN, MM>::supported, "MFMA sweep not available for this size");                       \
        static const int max_wv = route_int("DPILQR_MFMA_WAVES", 12);            \
        /* wavefronts per workgroup = per CU: 4 (one per SIMD), 8, or 12 when the launch has the items for them */  \
        const int wv = (bd && grid_items > 2048 && max_wv >= 12 && MfmaCfg<NN, MM>::total * 8 * 12 <= kMaxLds) ? 12 \
                       : ((grid_items > 1024 && max_wv >= 8) ? 8 : 4);                                              \
        g_sweep_waves = wv;                                                                                        \
        const size_t lds_t = sizeof(double) * MfmaCfg<NN, MM>::total * wv;                                        \
        auto kern = wv == 12 ? k_riccati_mfma<NN, MM, 12, 4, 2>                                                    \
                    : wv == 8 ? (bd ? k_riccati_mfma<NN, MM, 8, 4, 2> : k_riccati_mfma<NN, MM, 8, 0, 0>)           \
                              : (bd ? k_riccati_mfma<NN, MM, 4, 4, 2> : k_riccati_mfma<NN, MM, 4, 0, 0>);          \
        int32_t rc_t = allow_lds(kern, lds_t);                                                                     \
        if (rc_t) return rc_t;                                                                                     \
        /* whole rounds of one workgroup per CU; the kernel deals the live items over them (riccati_mfma.hpp) */    \
        const int cus = device_cus();                                                                              \
        const int grid = grid_items <= cus ? grid_items : (grid_items + cus * wv - 1) / (cus * wv) * cus;          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wv), lds_t, st, B, T, tiles, mu, K, d,                      \
                           singular, items, n_items, gains_by_item, cus, FusedArgs{});                             \
        HIP_TRY(hipGetLastError());                                                                                \
        return DPILQR_OK;                                                                                          \
    }
        DPILQR_TILED_SIZES(DPILQR_TRY_MFMA)
        // n_x = 24 (four six-state or six four-state agents): 19 KB of LDS per wavefront, two per SIMD; the workgroup sweep
        // costs 2.2 ms per 2048 items there, a wavefront per item 0.7 (profiles/r03_small_clusters.txt)
#undef DPILQR_TRY_MFMA
        if (n == 24 && m == 12) {   // the all-MFMA (dense) instantiation only: the block-diagonal lane mapping stops at five agents
            static_assert(MfmaCfg<24, 12>::supported, "MFMA sweep not available for this size");
            static const int max_wv = route_int("DPILQR_MFMA_WAVES", 8);
            const int wv = (grid_items > 1024 && max_wv >= 8) ? 8 : 4;
            g_sweep_waves = wv;
            const size_t lds_t = sizeof(double) * MfmaCfg<24, 12>::total * wv;
            auto kern = wv == 8 ? k_riccati_mfma<24, 12, 8, 0, 0> : k_riccati_mfma<24, 12, 4, 0, 0>;
            int32_t rc_t = allow_lds(kern, lds_t);
            if (rc_t) return rc_t;
            const int cus = device_cus();
            const int grid = grid_items <= cus ? grid_items : (grid_items + cus * wv - 1) / (cus * wv) * cus;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wv), lds_t, st, B, T, tiles, mu, K, d, singular, items, n_items,
                               gains_by_item, cus, FusedArgs{});
            HIP_TRY(hipGetLastError());
            return DPILQR_OK;
        }
    }
    // Cluster sizes without a wavefront instantiation of their own (n_x not a multiple of 4: three six-state agents,
    // CarDynamics3D; tiny: one six-state agent; a user plugin's odd sizes): the dense wavefront sweep of the next larger
    // instantiated size, which pads the records while loading them and stores the real block of the gains
    // (riccati_mfma.hpp, PAD).  Before: the workgroup sweep or the generic kernel -- slower than clusters twice the size
    // (profiles/r03_small_clusters.txt: three quadcopters 1.85 ms per 2048 items against 0.89 ms for four).
    static const bool no_pad = route_flag("DPILQR_RICCATI_NO_PAD");   // A/B switch
    // (not two twelve-state agents: n_u = 8 padded to 12 is slower than their workgroup sweep, 1.72 against 1.51 ms per 512 items)
    if (pick == 0 && !no_pad && n <= 24 && m <= 12 && !(block_ns == 12 && n == 24)) {
#define DPILQR_TRY_PAD(NN, MM)                                                                                     \
    if (n <= NN && m <= MM) {                                                                                      \
        static_assert(MfmaCfg<NN, MM>::supported, "MFMA sweep not available for this size");                       \
        static const int max_wv = route_int("DPILQR_MFMA_WAVES", 8);             \
        const int wv = (grid_items > 1024 && max_wv >= 8) ? 8 : 4;                                                 \
        g_sweep_waves = wv;                                                                                        \
        const size_t lds_t = sizeof(double) * MfmaCfg<NN, MM>::total * wv;                                        \
        auto kern = wv == 8 ? k_riccati_mfma_pad<NN, MM, 8> : k_riccati_mfma_pad<NN, MM, 4>;                       \
        int32_t rc_t = allow_lds(kern, lds_t);                                                                     \
        if (rc_t) return rc_t;                                                                                     \
        const int cus = device_cus();                                                                              \
        const int grid = grid_items <= cus ? grid_items : (grid_items + cus * wv - 1) / (cus * wv) * cus;          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wv), lds_t, st, B, T, tiles, mu, K, d, singular, items,      \
                           n_items, gains_by_item, cus, n, m);                                                     \
        HIP_TRY(hipGetLastError());                                                                                \
        return DPILQR_OK;                                                                                          \
    }
        DPILQR_TILED_SIZES(DPILQR_TRY_PAD)
        DPILQR_TRY_PAD(24, 12)
#undef DPILQR_TRY_PAD
    }
    // larger clusters of the library's own (block-diagonal) tiles: one workgroup per sub-problem, riccati_wg.hpp
    static const bool no_wg = route_flag("DPILQR_RICCATI_NO_WG");   // A/B switch
    if (pick == 0 && !no_wg && block_ns > 0) {
#define DPILQR_TRY_WG(KK, NS_, NC_)                                                                                 \
    if (block_ns == NS_ && block_nc == NC_ && n == KK * NS_ && m == KK * NC_) {                                     \
        using WC = WgCfg<KK * NS_, KK * NC_, NS_, NC_>;                                                             \
        static_assert(WC::supported, "workgroup sweep not available for this size");                                \
        const size_t lds_w = sizeof(double) * WC::total;                                                            \
        int32_t rc_w = allow_lds(k_riccati_wg<KK * NS_, KK * NC_, NS_, NC_>, lds_w);                                \
        if (rc_w) return rc_w;                                                                                      \
        hipLaunchKernelGGL((k_riccati_wg<KK * NS_, KK * NC_, NS_, NC_>), dim3(grid_items), dim3(kWgThreads), lds_w, \
                           st, B, T, tiles, mu, K, d, singular, items, n_items, gains_by_item, FusedArgs{});        \
        HIP_TRY(hipGetLastError());                                                                                 \
        return DPILQR_OK;                                                                                           \
    }
        // four-state models (DoubleInt4D, Unicycle4D), 6..15 agents; six-state models (DoubleInt6D, Quadcopter6D,
        // Human6D, HumanLin6D), 2..10 agents
        DPILQR_TRY_WG(6, 4, 2) DPILQR_TRY_WG(7, 4, 2) DPILQR_TRY_WG(8, 4, 2) DPILQR_TRY_WG(9, 4, 2) DPILQR_TRY_WG(10, 4, 2)
        DPILQR_TRY_WG(11, 4, 2) DPILQR_TRY_WG(12, 4, 2) DPILQR_TRY_WG(13, 4, 2) DPILQR_TRY_WG(14, 4, 2) DPILQR_TRY_WG(15, 4, 2)
        DPILQR_TRY_WG(2, 6, 3) DPILQR_TRY_WG(3, 6, 3) DPILQR_TRY_WG(4, 6, 3) DPILQR_TRY_WG(5, 6, 3) DPILQR_TRY_WG(6, 6, 3)
        DPILQR_TRY_WG(7, 6, 3) DPILQR_TRY_WG(8, 6, 3) DPILQR_TRY_WG(9, 6, 3) DPILQR_TRY_WG(10, 6, 3)
        // Quadcopter12D / the padded human, 2..5 agents
        DPILQR_TRY_WG(2, 12, 4) DPILQR_TRY_WG(3, 12, 4) DPILQR_TRY_WG(4, 12, 4) DPILQR_TRY_WG(5, 12, 4)
        // single six- and twelve-state agents (cfg4's k = 1 bucket, selfish_warmstart); CarDynamics3D pairs (n_x even)
        DPILQR_TRY_WG(1, 6, 3) DPILQR_TRY_WG(1, 12, 4)
        DPILQR_TRY_WG(2, 3, 2) DPILQR_TRY_WG(4, 3, 2) DPILQR_TRY_WG(6, 3, 2)
#undef DPILQR_TRY_WG
    }
    if (pick <= 1) {
#define DPILQR_TRY_TILED(NN, MM)                                                                                   \
    if (n == NN && m == MM) {                                                                                      \
        static_assert(TiledCfg<NN, MM>::supported, "tiled sweep not available for this size");                     \
        const size_t lds_t = sizeof(double) * TiledCfg<NN, MM>::total * kTiledWaves;                              \
        int32_t rc_t = allow_lds(k_riccati_tiled<NN, MM>, lds_t);                                                  \
        if (rc_t) return rc_t;                                                                                     \
        hipLaunchKernelGGL((k_riccati_tiled<NN, MM>), dim3((grid_items + kTiledWaves - 1) / kTiledWaves),          \
                           dim3(64 * kTiledWaves), lds_t, st, B, T, tiles, mu, K, d, singular, items, n_items,     \
                           gains_by_item);                                                                         \
        HIP_TRY(hipGetLastError());                                                                                \
        return DPILQR_OK;                                                                                          \
    }
        DPILQR_TILED_SIZES(DPILQR_TRY_TILED)
#undef DPILQR_TRY_TILED
    }
    const size_t lds = riccati_lds_bytes(n, m);
    int32_t rc = allow_lds(k_riccati_generic, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k_riccati_generic, dim3(grid_items), dim3(riccati_threads(n)), lds, st, B, T, n, m, tiles, mu, K,
                       d, singular, items, n_items, gains_by_item);
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

// The fused sweep (riccati_mfma.hpp, FUSED): no tile records; for batches of DoubleIntDynamics4D agents with one Q, R, Q_f
// for all agents and items and a planar proximity cost (the caller checks the descriptor's hints).  Returns
// DPILQR_EUNSUPPORTED without touching the error text when the shape has no fused instantiation.
int32_t launch_riccati_fused(const dpilqr_batch_desc& D, const double* X, const double* U, const double* mu, double* K,
                             double* d, int32_t* singular, const int32_t* items, const int32_t* n_items, int grid_items,
                             int gains_by_item, hipStream_t st) {
    g_sweep_waves = 0;
    if (grid_items <= 0) return DPILQR_OK;
    const int n = D.k * D.n_s, m = D.k * D.n_c;
    static const int max_wv = route_int("DPILQR_MFMA_WAVES", 12);
#define DPILQR_TRY_FUSED(NN, MM)                                                                                   \
    if (n == NN && m == MM) {                                                                                      \
        using CF = MfmaCfg<NN, MM, true>;                                                                          \
        const int wv = (grid_items > 2048 && max_wv >= 12 && CF::total * 8 * 12 <= kMaxLds) ? 12                   \
                       : ((grid_items > 1024 && max_wv >= 8) ? 8 : 4);                                              \
        g_sweep_waves = wv;                                                                                        \
        const size_t lds_t = sizeof(double) * CF::total * wv;                                                      \
        auto kern = wv == 12 ? k_riccati_mfma<NN, MM, 12, 4, 2, true>                                              \
                    : wv == 8 ? k_riccati_mfma<NN, MM, 8, 4, 2, true> : k_riccati_mfma<NN, MM, 4, 4, 2, true>;     \
        int32_t rc_t = allow_lds(kern, lds_t);                                                                     \
        if (rc_t) return rc_t;                                                                                     \
        const int cus = device_cus();                                                                              \
        const int grid = grid_items <= cus ? grid_items : (grid_items + cus * wv - 1) / (cus * wv) * cus;          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wv), lds_t, st, D.B, D.T, nullptr, mu, K, d, singular, items, \
                           n_items, gains_by_item, cus, FusedArgs{D, X, U});                                       \
        HIP_TRY(hipGetLastError());                                                                                \
        return DPILQR_OK;                                                                                          \
    }
    // the six-state family up to four agents, CarDynamics3D up to six: in-sweep production (tu_inprod.hip)
    {
        const int32_t rc_ip = launch_riccati_inprod(D, X, U, mu, K, d, singular, items, n_items, grid_items, gains_by_item, st);
        if (rc_ip != DPILQR_EUNSUPPORTED) return rc_ip;
    }
    // launches of at most one item per SIMD: a team of two wavefronts per item (tu_team.hip)
    if (grid_items <= 1024 && max_wv >= 4 && (fused_wavefront_sweep_applies(D) || fused_wavefront_general_applies(D))) {
        const int32_t rc_team = launch_riccati_team(D, X, U, mu, K, d, singular, items, n_items, grid_items, gains_by_item, st);
        if (rc_team != DPILQR_EUNSUPPORTED) {
            if (rc_team == DPILQR_OK) g_sweep_waves = 4;
            return rc_team;
        }
    }
    if (fused_wavefront_sweep_applies(D)) {
        DPILQR_TILED_SIZES(DPILQR_TRY_FUSED)
    }
#undef DPILQR_TRY_FUSED
    // the general form for the four-state family (FUSED == 2): UnicycleDynamics4D, per-agent / per-item weights
#define DPILQR_TRY_FUSED2(NN, MM)                                                                                  \
    if (n == NN && m == MM) {                                                                                      \
        using CF = MfmaCfg<NN, MM, 2>;                                                                             \
        const int wv = (grid_items > 1024 && max_wv >= 8) ? 8 : 4;                                                 \
        g_sweep_waves = wv;                                                                                        \
        const size_t lds_t = sizeof(double) * CF::total * wv;                                                      \
        auto kern = wv == 8 ? k_riccati_mfma_general<NN, MM, 8> : k_riccati_mfma_general<NN, MM, 4>;               \
        int32_t rc_t = allow_lds(kern, lds_t);                                                                     \
        if (rc_t) return rc_t;                                                                                     \
        const int cus = device_cus();                                                                              \
        const int grid = grid_items <= cus ? grid_items : (grid_items + cus * wv - 1) / (cus * wv) * cus;          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wv), lds_t, st, D.B, D.T, mu, K, d, singular, items,         \
                           n_items, gains_by_item, cus, FusedArgs{D, X, U});                                       \
        HIP_TRY(hipGetLastError());                                                                                \
        return DPILQR_OK;                                                                                          \
    }
    if (fused_wavefront_general_applies(D)) {
        DPILQR_TILED_SIZES(DPILQR_TRY_FUSED2)
    }
#undef DPILQR_TRY_FUSED2
    // larger clusters: the workgroup sweep, fused (riccati_wg.hpp): any models of the four- or six-state family
#define DPILQR_TRY_WGF(KK, NS_, NC_)                                                                                \
    if (D.n_s == NS_ && D.n_c == NC_ && D.k == KK) {                                                                \
        using WC = WgCfg<KK * NS_, KK * NC_, NS_, NC_, true>;                                                       \
        static_assert(WC::supported, "fused workgroup sweep not available for this size");                          \
        const size_t lds_w = sizeof(double) * WC::total;                                                            \
        int32_t rc_w = allow_lds(k_riccati_wg<KK * NS_, KK * NC_, NS_, NC_, true>, lds_w);                          \
        if (rc_w) return rc_w;                                                                                      \
        hipLaunchKernelGGL((k_riccati_wg<KK * NS_, KK * NC_, NS_, NC_, true>), dim3(grid_items), dim3(kWgThreads),  \
                           lds_w, st, D.B, D.T, nullptr, mu, K, d, singular, items, n_items, gains_by_item,         \
                           FusedArgs{D, X, U});                                                                     \
        HIP_TRY(hipGetLastError());                                                                                 \
        return DPILQR_OK;                                                                                           \
    }
    if (fused_workgroup_sweep_applies(D)) {
        DPILQR_TRY_WGF(6, 4, 2) DPILQR_TRY_WGF(7, 4, 2) DPILQR_TRY_WGF(8, 4, 2) DPILQR_TRY_WGF(9, 4, 2) DPILQR_TRY_WGF(10, 4, 2)
        DPILQR_TRY_WGF(11, 4, 2) DPILQR_TRY_WGF(12, 4, 2) DPILQR_TRY_WGF(13, 4, 2) DPILQR_TRY_WGF(14, 4, 2) DPILQR_TRY_WGF(15, 4, 2)
        DPILQR_TRY_WGF(2, 6, 3) DPILQR_TRY_WGF(3, 6, 3) DPILQR_TRY_WGF(4, 6, 3) DPILQR_TRY_WGF(5, 6, 3) DPILQR_TRY_WGF(6, 6, 3)
        DPILQR_TRY_WGF(7, 6, 3) DPILQR_TRY_WGF(8, 6, 3) DPILQR_TRY_WGF(9, 6, 3) DPILQR_TRY_WGF(10, 6, 3)
    }
#undef DPILQR_TRY_WGF
    return DPILQR_EUNSUPPORTED;
}

int32_t set_stamp_buffer_riccati(void* buf) {
    void* p = buf;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)));
    return DPILQR_OK;
}

}  // namespace dpilqr
