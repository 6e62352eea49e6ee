// dpilqr_hip.hip -- C ABI (include/dpilqr_hip.h) over the gfx950 kernels.
//
// Host side of libdpilqr_hip.so: argument validation, kernel dispatch by per-agent dimension family,
// and the device-resident iLQR iteration loop (ilqrSolver.solve, control.py:150-225).
// Built by __graft_entry__.build():  hipcc --offload-arch=gfx950 -O3 -shared -fPIC ...
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dpilqr_hip.h"
#include "forward.hpp"
#include "models.hpp"
#include "riccati.hpp"
#include "riccati_mfma.hpp"
#include "forward_wave.hpp"
#include "riccati_wg.hpp"
#include "tiles_wave.hpp"
#include "riccati_tiled.hpp"
#include "tiles.hpp"

using namespace dpilqr;

namespace {

thread_local char g_err[512] = "";

int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail(DPILQR_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int kMaxLds = 160 * 1024;  // gfx950: 160 KiB per workgroup
constexpr int kMaxAgents = 64;

int family_nc(int ns) { return ns == 3 ? 2 : ns == 4 ? 2 : ns == 6 ? 3 : ns == 12 ? 4 : -1; }

int32_t check_desc(const dpilqr_batch_desc* d) {
    if (!d) return fail(DPILQR_EINVAL, "desc is NULL");
    if (d->B < 0 || d->k < 1 || d->T < 1) return fail(DPILQR_EINVAL, "bad sizes B=%d k=%d T=%d", d->B, d->k, d->T);
    if (d->k > kMaxAgents) return fail(DPILQR_EUNSUPPORTED, "k=%d agents per sub-problem exceeds %d", d->k, kMaxAgents);
    if (family_nc(d->n_s) != d->n_c)
        return fail(DPILQR_EINVAL, "(n_s,n_c)=(%d,%d) is not a model family; expected (3,2),(4,2),(6,3),(12,4)", d->n_s, d->n_c);
    if (d->B > 0 && (!d->model || !d->n_dims || !d->xf || !d->Q || !d->R || !d->Qf || !d->radius))   // an empty batch owns nothing
        return fail(DPILQR_EINVAL, "desc holds a NULL device pointer");
    return DPILQR_OK;
}

hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// compute units of the current device (256 on MI355X): the sweep deals its items over rounds of this many workgroups
int device_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    return cus;
}

// hints packed into dpilqr_batch_desc::uniform_model (include/dpilqr_hip.h): -1 = unknown / mixed
inline int hint_model(const dpilqr_batch_desc& D) { return (D.uniform_model & 0xff) - 1; }
inline int hint_n_dims(const dpilqr_batch_desc& D) { return ((D.uniform_model >> 8) & 0xff) - 1; }

template <typename Kern>
int32_t allow_lds(Kern kern, size_t bytes) {
    if (bytes > (size_t)kMaxLds) return fail(DPILQR_EUNSUPPORTED, "needs %zu B of LDS per workgroup (> %d)", bytes, kMaxLds);
    if (bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return DPILQR_OK;
}

// run `body` with the (NS,NC) family as compile-time constants
#define DISPATCH_FAMILY(ns, BODY)                                                         \
    switch (ns) {                                                                         \
    case 3:  { constexpr int NS = 3,  NC = 2; BODY } break;                               \
    case 4:  { constexpr int NS = 4,  NC = 2; BODY } break;                               \
    case 6:  { constexpr int NS = 6,  NC = 3; BODY } break;                               \
    case 12: { constexpr int NS = 12, NC = 4; BODY } break;                               \
    default: return fail(DPILQR_EINVAL, "unsupported per-agent state dim %d", (int)(ns)); \
    }

int riccati_threads(int n) { return n <= 24 ? 64 : (n <= 36 ? 128 : 256); }

int forward_threads(int k, int ngrp) {
    const int t = ((k * ngrp + 63) / 64) * 64;
    return t;
}

int32_t launch_make_tiles(const dpilqr_batch_desc& D, const double* X, const double* U, double* tiles,
                          const int32_t* items, const int32_t* n_items, int grid_items, bool sparse, bool dyn_only,
                          hipStream_t st) {
    if (grid_items <= 0) return DPILQR_OK;
    static const bool force_dense = getenv("DPILQR_TILES_DENSE") != nullptr;   // A/B switch
    if (force_dense) sparse = false;
    if (!sparse) dyn_only = false;
    // the solve loop's producer for a batch of one linear model whose (X, U)-independent entries are already in place:
    // kernels compiled per (model, agents), tiles_wave.hpp.  (Measured: for the other cases -- A, B to be written too,
    // or more than 6 agents -- the generic producer's sparse stores are the faster ones.)
    static const bool no_wave = getenv("DPILQR_TILES_GENERIC") != nullptr;   // A/B switch
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

// compile-time-sized sweeps (one wavefront per sub-problem); everything else takes the generic kernel
#define DPILQR_TILED_SIZES(X) X(4, 2) X(8, 4) X(12, 6) X(16, 8) X(20, 10)

thread_local int g_sweep_waves = 0;   // wavefronts per workgroup of the last launch_riccati (0: not the wavefront sweep)

int32_t launch_riccati(int B, int T, int n, int m, const double* tiles, const double* mu, double* K, double* d,
                       int32_t* singular, const int32_t* items, const int32_t* n_items, int grid_items,
                       int gains_by_item, int block_ns, int block_nc, hipStream_t st) {
    g_sweep_waves = 0;
    if (grid_items <= 0) return DPILQR_OK;
    // block_ns > 0: the caller guarantees that [A|B] is block diagonal with block_ns x (block_ns + block_nc) blocks
    // (tiles made by k_make_tiles from a MultiDynamicalModel); 0: arbitrary dense tiles (the plugin boundary).
    static const bool no_bd = getenv("DPILQR_RICCATI_DENSE") != nullptr;   // A/B switch
    const bool bd = !no_bd && block_ns == 4 && block_nc == 2 && n == 4 * (m / 2) && m % 2 == 0;
    // sweep selection: matrix-pipe kernel where instantiated, else the vector-pipe tiled kernel, else the generic one
    // (DPILQR_RICCATI=mfma|tiled|generic pins one for A/B measurements)
    static const char* pick_env = getenv("DPILQR_RICCATI");
    static const int pick = getenv("DPILQR_FORCE_GENERIC_RICCATI") ? 2
                            : (!pick_env ? 0 : (!strcmp(pick_env, "tiled") ? 1 : (!strcmp(pick_env, "generic") ? 2 : 0)));
    if (pick == 0) {
#define DPILQR_TRY_MFMA(NN, MM)                                                                                    \
    if (n == NN && m == MM) {                                                                                      \
        static_assert(MfmaCfg<NN, MM>::supported, "MFMA sweep not available for this size");                       \
        static const int max_wv = getenv("DPILQR_MFMA_WAVES") ? atoi(getenv("DPILQR_MFMA_WAVES")) : 12;            \
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
                           singular, items, n_items, gains_by_item, cus);                                          \
        HIP_TRY(hipGetLastError());                                                                                \
        return DPILQR_OK;                                                                                          \
    }
        DPILQR_TILED_SIZES(DPILQR_TRY_MFMA)
#undef DPILQR_TRY_MFMA
    }
    // larger clusters of the library's own (block-diagonal) tiles: one workgroup per sub-problem, riccati_wg.hpp
    static const bool no_wg = getenv("DPILQR_RICCATI_NO_WG") != nullptr;   // A/B switch
    if (pick == 0 && !no_wg && block_ns > 0) {
#define DPILQR_TRY_WG(KK, NS_, NC_)                                                                                 \
    if (block_ns == NS_ && block_nc == NC_ && n == KK * NS_ && m == KK * NC_) {                                     \
        using WC = WgCfg<KK * NS_, KK * NC_, NS_, NC_>;                                                             \
        static_assert(WC::supported, "workgroup sweep not available for this size");                                \
        const size_t lds_w = sizeof(double) * WC::total;                                                            \
        int32_t rc_w = allow_lds(k_riccati_wg<KK * NS_, KK * NC_, NS_, NC_>, lds_w);                                \
        if (rc_w) return rc_w;                                                                                      \
        hipLaunchKernelGGL((k_riccati_wg<KK * NS_, KK * NC_, NS_, NC_>), dim3(grid_items), dim3(kWgThreads), lds_w, \
                           st, B, T, tiles, mu, K, d, singular, items, n_items, gains_by_item);                     \
        HIP_TRY(hipGetLastError());                                                                                 \
        return DPILQR_OK;                                                                                           \
    }
        // four-state models (DoubleInt4D, Unicycle4D), 6..15 agents; six-state models (DoubleInt6D, Quadcopter6D,
        // Human6D, HumanLin6D), 2..10 agents
        DPILQR_TRY_WG(6, 4, 2) DPILQR_TRY_WG(7, 4, 2) DPILQR_TRY_WG(8, 4, 2) DPILQR_TRY_WG(9, 4, 2) DPILQR_TRY_WG(10, 4, 2)
        DPILQR_TRY_WG(11, 4, 2) DPILQR_TRY_WG(12, 4, 2) DPILQR_TRY_WG(13, 4, 2) DPILQR_TRY_WG(14, 4, 2) DPILQR_TRY_WG(15, 4, 2)
        DPILQR_TRY_WG(2, 6, 3) DPILQR_TRY_WG(3, 6, 3) DPILQR_TRY_WG(4, 6, 3) DPILQR_TRY_WG(5, 6, 3) DPILQR_TRY_WG(6, 6, 3)
        DPILQR_TRY_WG(7, 6, 3) DPILQR_TRY_WG(8, 6, 3) DPILQR_TRY_WG(9, 6, 3) DPILQR_TRY_WG(10, 6, 3)
        // Quadcopter12D, 2..5 agents
        DPILQR_TRY_WG(2, 12, 4) DPILQR_TRY_WG(3, 12, 4) DPILQR_TRY_WG(4, 12, 4) DPILQR_TRY_WG(5, 12, 4)
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

static bool no_wave_ro() { static const bool v = getenv("DPILQR_FORWARD_GENERIC") != nullptr; return v; }

int32_t launch_forward(const dpilqr_batch_desc& D, int mode, const double* x0, double* X, double* U, const double* K,
                       const double* d, const double* alphas, int ngrp, double* Xc, double* Uc, double* Jc,
                       const SolveState& S, const int32_t* items, const int32_t* n_items, int grid_items,
                       hipStream_t st) {
    if (grid_items <= 0) return DPILQR_OK;
    const int n = D.k * D.n_s, m = D.k * D.n_c;
    int threads = forward_threads(D.k, mode == kModeRollout ? 1 : ngrp);
    if (threads > 256) return fail(DPILQR_EUNSUPPORTED, "k*n_alpha=%d exceeds the 256-thread workgroup of the forward pass", D.k * ngrp);
    const size_t lds_item = (forward_lds_bytes(n, m, D.k, ngrp) + 15) & ~(size_t)15;
    if (mode != kModeRollout && (m * n + threads - 1) / threads > kMaxStage)
        return fail(DPILQR_EUNSUPPORTED, "n_u*n_x=%d exceeds the forward pass's per-step staging (%d threads x %d)", m * n, threads, kMaxStage);
    if (!no_wave_ro() && mode == kModeRollout && hint_model(D) >= 0 && !items) {
        const int model = hint_model(D);
#define DPILQR_TRY_RO(MODEL, KA)                                                                                    \
    if (model == MODEL && D.k == KA && model_ns(MODEL) == D.n_s && model_nc(MODEL) == D.n_c) {                      \
        using WR = WaveRolloutLds<MODEL, KA>;                                                                       \
        const int per_wg = 4 * WR::IPW;                                                                             \
        hipLaunchKernelGGL((k_rollout_wave<MODEL, KA>), dim3((grid_items + per_wg - 1) / per_wg), dim3(256),        \
                           sizeof(double) * WR::total * 4, st, D, x0, U, X, Jc);                                    \
        HIP_TRY(hipGetLastError());                                                                                 \
        return DPILQR_OK;                                                                                           \
    }
#define DPILQR_RO_10(MODEL) DPILQR_TRY_RO(MODEL, 1) DPILQR_TRY_RO(MODEL, 2) DPILQR_TRY_RO(MODEL, 3)                \
        DPILQR_TRY_RO(MODEL, 4) DPILQR_TRY_RO(MODEL, 5) DPILQR_TRY_RO(MODEL, 6) DPILQR_TRY_RO(MODEL, 7)             \
        DPILQR_TRY_RO(MODEL, 8) DPILQR_TRY_RO(MODEL, 9) DPILQR_TRY_RO(MODEL, 10)
#define DPILQR_RO_15(MODEL) DPILQR_RO_10(MODEL) DPILQR_TRY_RO(MODEL, 11) DPILQR_TRY_RO(MODEL, 12)                   \
        DPILQR_TRY_RO(MODEL, 13) DPILQR_TRY_RO(MODEL, 14) DPILQR_TRY_RO(MODEL, 15)
        DPILQR_RO_15(kDoubleInt4D)
        DPILQR_RO_15(kUnicycle4D)
        DPILQR_RO_10(kQuadcopter6D)
#define DPILQR_RO_6(MODEL) DPILQR_TRY_RO(MODEL, 1) DPILQR_TRY_RO(MODEL, 2) DPILQR_TRY_RO(MODEL, 3)                 \
        DPILQR_TRY_RO(MODEL, 4) DPILQR_TRY_RO(MODEL, 5) DPILQR_TRY_RO(MODEL, 6)
        DPILQR_RO_6(kDoubleInt6D)
        DPILQR_RO_6(kCar3D)
        DPILQR_RO_6(kHuman6D)
        DPILQR_RO_6(kHumanLin6D)
        DPILQR_TRY_RO(kQuadcopter12D, 1) DPILQR_TRY_RO(kQuadcopter12D, 2) DPILQR_TRY_RO(kQuadcopter12D, 3)
        DPILQR_TRY_RO(kQuadcopter12D, 4) DPILQR_TRY_RO(kQuadcopter12D, 5)
#undef DPILQR_RO_6
#undef DPILQR_RO_15
#undef DPILQR_RO_10
#undef DPILQR_TRY_RO
    }
    // one solver iteration's line search for a batch of ONE model whose candidates fit a wavefront: the kernels
    // compiled for (model, agents), see forward_wave.hpp
    static const bool no_wave = getenv("DPILQR_FORWARD_GENERIC") != nullptr;   // A/B switch
    if (!no_wave && mode == kModeLineSearch && hint_model(D) >= 0 && ngrp == DPILQR_N_ALPHA && items && n_items) {
        const int model = hint_model(D);
#define DPILQR_TRY_WAVE(MODEL, KA)                                                                                  \
    if (model == MODEL && D.k == KA && model_ns(MODEL) == D.n_s && model_nc(MODEL) == D.n_c) {                      \
        using WF = WaveFwdLds<MODEL, KA>;                                                                           \
        const size_t lds_w = sizeof(double) * WF::total * WF::IPB;                                                  \
        int32_t rc_w = allow_lds(k_linesearch_wave<MODEL, KA>, lds_w);                                              \
        if (rc_w) return rc_w;                                                                                      \
        hipLaunchKernelGGL((k_linesearch_wave<MODEL, KA>), dim3((grid_items + WF::IPB - 1) / WF::IPB),              \
                           dim3(64 * WF::NW * WF::IPB), lds_w, st, D, X, U, K, d, alphas, Xc, Uc, S, items, n_items); \
        HIP_TRY(hipGetLastError());                                                                                 \
        return DPILQR_OK;                                                                                           \
    }
#define DPILQR_WAVE_10(MODEL) DPILQR_TRY_WAVE(MODEL, 1) DPILQR_TRY_WAVE(MODEL, 2) DPILQR_TRY_WAVE(MODEL, 3)        \
        DPILQR_TRY_WAVE(MODEL, 4) DPILQR_TRY_WAVE(MODEL, 5) DPILQR_TRY_WAVE(MODEL, 6) DPILQR_TRY_WAVE(MODEL, 7)     \
        DPILQR_TRY_WAVE(MODEL, 8) DPILQR_TRY_WAVE(MODEL, 9) DPILQR_TRY_WAVE(MODEL, 10)
#define DPILQR_WAVE_15(MODEL) DPILQR_WAVE_10(MODEL) DPILQR_TRY_WAVE(MODEL, 11) DPILQR_TRY_WAVE(MODEL, 12)           \
        DPILQR_TRY_WAVE(MODEL, 13) DPILQR_TRY_WAVE(MODEL, 14) DPILQR_TRY_WAVE(MODEL, 15)
        DPILQR_WAVE_15(kDoubleInt4D)
        DPILQR_WAVE_15(kUnicycle4D)
        DPILQR_WAVE_10(kQuadcopter6D)
        // the remaining models, up to six agents (one wavefront per sub-problem)
#define DPILQR_WAVE_6(MODEL) DPILQR_TRY_WAVE(MODEL, 1) DPILQR_TRY_WAVE(MODEL, 2) DPILQR_TRY_WAVE(MODEL, 3)         \
        DPILQR_TRY_WAVE(MODEL, 4) DPILQR_TRY_WAVE(MODEL, 5) DPILQR_TRY_WAVE(MODEL, 6)
        DPILQR_WAVE_6(kDoubleInt6D)
        DPILQR_WAVE_6(kCar3D)
        DPILQR_WAVE_6(kHuman6D)
        DPILQR_WAVE_6(kHumanLin6D)
        DPILQR_TRY_WAVE(kQuadcopter12D, 1) DPILQR_TRY_WAVE(kQuadcopter12D, 2) DPILQR_TRY_WAVE(kQuadcopter12D, 3)
        DPILQR_TRY_WAVE(kQuadcopter12D, 4) DPILQR_TRY_WAVE(kQuadcopter12D, 5)
#undef DPILQR_WAVE_6
#undef DPILQR_WAVE_15
#undef DPILQR_WAVE_10
#undef DPILQR_TRY_WAVE
    }
    // single-wave sub-problems are packed four to a workgroup (one wave per SIMD), see forward.hpp
    static const bool no_pack = getenv("DPILQR_FORWARD_NO_PACK") != nullptr;   // diagnostic switch
    const int ipb = (!no_pack && threads == 64 && 4 * lds_item <= (size_t)kMaxLds) ? 4 : 1;
    const size_t lds = lds_item * ipb;
    threads *= ipb;
    DISPATCH_FAMILY(D.n_s, {
        int32_t rc = allow_lds(k_forward<NS, NC>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((k_forward<NS, NC>), dim3((grid_items + ipb - 1) / ipb), dim3(threads), lds, st, D, mode, x0, X, U,
                           K, d, alphas, ngrp, Xc, Uc, Jc, S, items, n_items, ipb, (int)(lds_item / sizeof(double)));
    })
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

// float32-rounded table of control.py:162 (quirk Q1), bit patterns of 1.1 ** (-arange(10, f32) ** 2)
void alpha_table(double* a) {
    static const uint32_t bits[DPILQR_N_ALPHA] = {0x3f800000u, 0x3f68ba2eu, 0x3f2ed9f7u, 0x3ed92350u, 0x3e5eda27u,
                                                  0x3dbd05a8u, 0x3d04808du, 0x3c19864au, 0x3b13029cu, 0x39e8ae70u};
    for (int i = 0; i < DPILQR_N_ALPHA; ++i) {
        float f;
        memcpy(&f, &bits[i], 4);
        a[i] = (double)f;
    }
}

__global__ void k_init_state(int B, double* mu, double* delta, int32_t* status, int32_t* n_bwd, int32_t* n_fwd,
                             int32_t* singular, int32_t* counts, int n_counts, double* alphas_dev, double a0, double a1,
                             double a2, double a3, double a4, double a5, double a6, double a7, double a8, double a9) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {  // _reset_regularization, control.py:227-230
        mu[i] = 1.0; delta[i] = 2.0; status[i] = DPILQR_STATUS_ACTIVE; n_bwd[i] = 0; n_fwd[i] = 0; singular[i] = 0;
    }
    if (i < n_counts) counts[i] = 0;
    if (i == 0) {
        alphas_dev[0] = a0; alphas_dev[1] = a1; alphas_dev[2] = a2; alphas_dev[3] = a3; alphas_dev[4] = a4;
        alphas_dev[5] = a5; alphas_dev[6] = a6; alphas_dev[7] = a7; alphas_dev[8] = a8; alphas_dev[9] = a9;
    }
}

// Admission, decided on the device where the exact number of survivors is known: behind the survivors
// that the previous iteration's line search pushed, append as many not-yet-started items as fit in the
// window, and clear the counter the NEXT iteration will push into.  ctl = {count, admitted} mailbox copy.
__global__ void k_admit(int32_t* list, int32_t* count, int32_t* next_count, int32_t* admitted, int B, int window,
                        int32_t* mail) {
    const int base = *count, first = *admitted;
    const int n_new = min(B - first, window - base);
    __syncthreads();
    for (int i = threadIdx.x; i < n_new; i += blockDim.x) list[base + i] = first + i;
    if (threadIdx.x == 0) {
        *count = base + n_new; *admitted = first + n_new; *next_count = 0;
        mail[0] = base + n_new; mail[1] = first + n_new;   // pinned host memory: the host reads it after the event
    }
}

__global__ void k_copy_f64(int n, const double* src, double* dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ void k_finish_status(int B, int32_t* status) {  // n_lqr_iter == 0: nothing ran
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B && status[i] == DPILQR_STATUS_ACTIVE) status[i] = DPILQR_STATUS_MAX_ITER;
}

constexpr int kCountRing = 4;

struct SolveWorkspace {
    // W = window = most sub-problems in flight at once: the big per-iteration buffers (tile records, gains,
    // line-search candidates) are indexed by position in the active list and sized by W, not by B.
    size_t tiles, K, d, Xc, Uc, mu, delta, J_star, J_last, alphas, singular, lists, counts, total;
    SolveWorkspace(const dpilqr_batch_desc& D, int W, bool gains_in_ws) {
        const size_t B = D.B, n = (size_t)D.k * D.n_s, m = (size_t)D.k * D.n_c, T = D.T, Wn = W;
        const TileLayout L((int)n, (int)m);
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        size_t o = 0;
        tiles = o;    o = al(o + sizeof(double) * Wn * (T + 1) * L.stride);
        K = o;        o = al(o + (gains_in_ws ? sizeof(double) * Wn * T * m * n : 0));
        d = o;        o = al(o + (gains_in_ws ? sizeof(double) * Wn * T * m : 0));
        // line-search candidates: every alpha's trajectory, so that accepting one is a copy, not a re-roll
        Xc = o;       o = al(o + sizeof(double) * Wn * DPILQR_N_ALPHA * (T + 1) * n);
        Uc = o;       o = al(o + sizeof(double) * Wn * DPILQR_N_ALPHA * T * m);
        mu = o;       o = al(o + sizeof(double) * B);
        delta = o;    o = al(o + sizeof(double) * B);
        J_star = o;   o = al(o + sizeof(double) * B);
        J_last = o;   o = al(o + sizeof(double) * B);
        alphas = o;   o = al(o + sizeof(double) * DPILQR_N_ALPHA);
        singular = o; o = al(o + sizeof(int32_t) * B);
        lists = o;    o = al(o + sizeof(int32_t) * 2 * Wn);
        counts = o;   o = al(o + sizeof(int32_t) * (kCountRing + 1));   // ring + the `admitted` counter
        total = o;
    }
};
constexpr int kMaxLqrIter = 4096;
constexpr int kMaxGlobalIter = 1 << 20;  // launches of the iteration loop one solve_batch call may make

int window_of(const dpilqr_batch_desc& D, int window) { return (window <= 0 || window > D.B) ? (D.B > 0 ? D.B : 1) : window; }

// The host reads the device-side counters kHostLag iterations late: that many iterations of launches are always
// queued behind the one the GPU is running, so a host thread that loses its core for a millisecond (a loaded box)
// does not leave the GPU idle.  The price is kHostLag empty iterations (a dozen tiny launches) at the end of a solve.
constexpr int kHostLag = 3, kMailRing = kHostLag + 1;

struct Mailbox {  // pinned host words the admission kernel posts the active-list counters into
    int32_t* host = nullptr;
    int32_t* dev = nullptr;   // the same words as the device sees them
    std::vector<int32_t> hist;  // exact active-list length of every global iteration of the last solve
    hipEvent_t ev[kMailRing] = {};
    ~Mailbox() {
        if (host) (void)hipHostFree(host);
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};
thread_local Mailbox g_mail;

// opt-in per-kernel timing (dpilqr_profile_*): event pairs recorded on the solve's own stream
struct Profiler {
    bool on = false;
    int mask = 0xF;   // classes that get events (0 tiles, 1 riccati, 2 forward, 3 rollout); every event pair costs a
                      // dispatch gap, so a caller that needs one kernel's duration asks for that one only
    bool skip = false;
    double ms[4] = {0, 0, 0, 0};
    int64_t launches[4] = {0, 0, 0, 0}, items[4] = {0, 0, 0, 0};
    // the wavefront sweep's launches by variant (4, 8, 12 wavefronts per workgroup -> index 0, 1, 2)
    double sweep_ms[3] = {0, 0, 0};
    int64_t sweep_launches[3] = {0, 0, 0}, sweep_items[3] = {0, 0, 0};
    std::vector<hipEvent_t> pool;
    struct Rec { int cls, iter; size_t e0; int tag; };
    std::vector<Rec> recs;
    size_t used = 0;
    hipEvent_t next() {
        if (used == pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[used++];
    }
    void begin(int cls, int iter, hipStream_t st) {
        skip = !on || !((mask >> cls) & 1);
        if (skip) return;
        recs.push_back({cls, iter, used, 0});
        hipEvent_t e = next();
        if (e) (void)hipEventRecord(e, st);
    }
    void end(hipStream_t st, int tag = 0) {   // tag: which variant ran (the sweep: wavefronts per workgroup, else 0)
        if (skip) return;
        hipEvent_t e = next();
        if (e) (void)hipEventRecord(e, st);
        recs.back().tag = tag;
    }
    // after the stream has been synchronised; active[it] = items processed by iteration it
    void collect(const std::vector<int32_t>& active, int B) {
        if (!on) return;
        for (const Rec& r : recs) {
            float t = 0.f;
            if (r.e0 + 1 < pool.size() && hipEventElapsedTime(&t, pool[r.e0], pool[r.e0 + 1]) == hipSuccess) {
                const int64_t n_it = (r.iter < 0) ? B : (r.iter < (int)active.size() ? active[r.iter] : 0);
                ms[r.cls] += t;
                launches[r.cls] += 1;
                items[r.cls] += n_it;
                if (r.cls == 1 && (r.tag == 4 || r.tag == 8 || r.tag == 12)) {
                    const int v = r.tag / 4 - 1;
                    sweep_ms[v] += t; sweep_launches[v] += 1; sweep_items[v] += n_it;
                }
            }
        }
        recs.clear();
        used = 0;
    }
};
thread_local Profiler g_prof;

}  // namespace

extern "C" {

int32_t dpilqr_abi_version(void) { return DPILQR_ABI_VERSION; }
const char* dpilqr_last_error(void) { return g_err; }

int32_t dpilqr_device_info(int32_t dev, int32_t* n_cu, int32_t* lds_bytes, char* arch, int32_t arch_len) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(DPILQR_ENOGPU, "no HIP device visible");
    if (dev < 0 || dev >= count) return fail(DPILQR_EINVAL, "device %d out of range (%d visible)", dev, count);
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int32_t)p.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return fail(DPILQR_ENOGPU, "device %d is %s; this library is built for gfx950 only", dev, p.gcnArchName);
    return DPILQR_OK;
}

int32_t dpilqr_model_dims(int32_t model, int32_t* n_s, int32_t* n_c) {
    if (model < 0 || model >= kNumModels || !n_s || !n_c) return fail(DPILQR_EINVAL, "unknown model %d", model);
    *n_s = model_ns(model);
    *n_c = model_nc(model);
    return DPILQR_OK;
}

static int32_t model_op(int op, int32_t n, int32_t ns, const int32_t* model, const double* x, const double* u, double dt,
                        double* o1, double* o2, void* stream) {
    if (n < 0 || !model || !x || !u || !o1 || (op == 2 && !o2)) return fail(DPILQR_EINVAL, "model op: bad argument");
    if (n == 0) return DPILQR_OK;
    const dim3 grid((n + 63) / 64), block(64);
    DISPATCH_FAMILY(ns, {
        if (op == 0) hipLaunchKernelGGL((k_model_op<NS, NC, 0>), grid, block, 0, as_stream(stream), n, model, x, u, dt, o1, o2);
        else if (op == 1) hipLaunchKernelGGL((k_model_op<NS, NC, 1>), grid, block, 0, as_stream(stream), n, model, x, u, dt, o1, o2);
        else hipLaunchKernelGGL((k_model_op<NS, NC, 2>), grid, block, 0, as_stream(stream), n, model, x, u, dt, o1, o2);
    })
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t dpilqr_model_f(int32_t n, int32_t family_ns, const int32_t* model, const double* x, const double* u,
                       double* x_dot, void* stream) {
    return model_op(0, n, family_ns, model, x, u, 0.0, x_dot, nullptr, stream);
}
int32_t dpilqr_model_integrate(int32_t n, int32_t family_ns, const int32_t* model, const double* x, const double* u,
                               double dt, double* x_new, void* stream) {
    return model_op(1, n, family_ns, model, x, u, dt, x_new, nullptr, stream);
}
int32_t dpilqr_model_linearize(int32_t n, int32_t family_ns, const int32_t* model, const double* x, const double* u,
                               double dt, double* A, double* B, void* stream) {
    return model_op(2, n, family_ns, model, x, u, dt, A, B, stream);
}

int32_t dpilqr_cost_eval(const dpilqr_batch_desc* desc, int32_t n_pts, const double* x, const double* u,
                         int32_t terminal, double* cost, void* stream) {
    int32_t rc = check_desc(desc);
    if (rc) return rc;
    if (n_pts < 0 || !x || !u || !cost) return fail(DPILQR_EINVAL, "cost_eval: bad argument");
    const int64_t total = (int64_t)desc->B * n_pts;
    if (total == 0) return DPILQR_OK;
    const dim3 grid((unsigned)((total + 63) / 64)), block(64);
    DISPATCH_FAMILY(desc->n_s, {
        hipLaunchKernelGGL((k_cost_eval<NS, NC>), grid, block, 0, as_stream(stream), *desc, n_pts, x, u, terminal, cost);
    })
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t dpilqr_tile_layout(int32_t n_x, int32_t n_u, int64_t offsets[7], int64_t row_strides[7], int64_t* stride) {
    if (n_x < 1 || n_u < 1 || !offsets || !row_strides || !stride) return fail(DPILQR_EINVAL, "tile_layout: bad argument");
    const TileLayout L(n_x, n_u);
    offsets[0] = L.oA; offsets[1] = L.oB; offsets[2] = L.oLxx; offsets[3] = L.oLux; offsets[4] = L.oLuu;
    offsets[5] = L.oLx; offsets[6] = L.oLu;
    row_strides[0] = L.ldAB; row_strides[1] = L.ldAB; row_strides[2] = n_x; row_strides[3] = L.ldUG; row_strides[4] = L.ldUG;
    row_strides[5] = 1; row_strides[6] = 1;
    *stride = L.stride;
    return DPILQR_OK;
}

int64_t dpilqr_tiles_bytes(int32_t B, int32_t T, int32_t n_x, int32_t n_u) {
    if (B < 0 || T < 1 || n_x < 1 || n_u < 1) return fail(DPILQR_EINVAL, "tiles_bytes: bad argument");
    return (int64_t)sizeof(double) * B * (T + 1) * TileLayout(n_x, n_u).stride;
}

int32_t dpilqr_make_tiles(const dpilqr_batch_desc* desc, const double* X, const double* U, double* tiles,
                          const int32_t* items, const int32_t* n_items, void* stream) {
    int32_t rc = check_desc(desc);
    if (rc) return rc;
    if (!X || !U || !tiles) return fail(DPILQR_EINVAL, "make_tiles: NULL pointer");
    return launch_make_tiles(*desc, X, U, tiles, items, n_items, desc->B, false, false, as_stream(stream));
}

int32_t dpilqr_rollout(const dpilqr_batch_desc* desc, const double* x0, const double* U, double* X, double* J,
                       void* stream) {
    int32_t rc = check_desc(desc);
    if (rc) return rc;
    if (desc->B == 0) return DPILQR_OK;
    if (!x0 || !U || !X || !J) return fail(DPILQR_EINVAL, "rollout: NULL pointer");
    SolveState S{};
    return launch_forward(*desc, kModeRollout, x0, X, const_cast<double*>(U), nullptr, nullptr, nullptr, 1, nullptr,
                          nullptr, J, S, nullptr, nullptr, desc->B, as_stream(stream));
}

int32_t dpilqr_backward_pass_tiles(int32_t B, int32_t T, int32_t n_x, int32_t n_u, const double* tiles,
                                   const double* mu, double* K, double* d, int32_t* singular, const int32_t* items,
                                   const int32_t* n_items, void* stream) {
    if (B < 0 || T < 1 || n_x < 1 || n_u < 1) return fail(DPILQR_EINVAL, "backward_pass_tiles: bad sizes");
    if (!tiles || !mu || !K || !d) return fail(DPILQR_EINVAL, "backward_pass_tiles: NULL pointer");
    return launch_riccati(B, T, n_x, n_u, tiles, mu, K, d, singular, items, n_items, B, 0, 0, 0, as_stream(stream));
}

int32_t dpilqr_backward_pass_tiles_blocks(int32_t B, int32_t T, int32_t n_x, int32_t n_u, int32_t block_ns,
                                          int32_t block_nc, const double* tiles, const double* mu, double* K,
                                          double* d, int32_t* singular, const int32_t* items,
                                          const int32_t* n_items, void* stream) {
    if (B < 0 || T < 1 || n_x < 1 || n_u < 1) return fail(DPILQR_EINVAL, "backward_pass_tiles_blocks: bad sizes");
    if (!tiles || !mu || !K || !d) return fail(DPILQR_EINVAL, "backward_pass_tiles_blocks: NULL pointer");
    if (block_ns < 0 || block_nc < 0 || (block_ns > 0 && (block_nc < 1 || n_x % block_ns || n_u % block_nc ||
                                                          n_x / block_ns != n_u / block_nc)))
        return fail(DPILQR_EINVAL, "backward_pass_tiles_blocks: n_x=%d, n_u=%d are not k blocks of %d, %d", n_x, n_u,
                    block_ns, block_nc);
    return launch_riccati(B, T, n_x, n_u, tiles, mu, K, d, singular, items, n_items, B, 0, block_ns, block_nc,
                          as_stream(stream));
}

int32_t dpilqr_backward_pass(const dpilqr_batch_desc* desc, const double* X, const double* U, const double* mu,
                             double* K, double* d, double* tiles_workspace, void* stream) {
    int32_t rc = check_desc(desc);
    if (rc) return rc;
    if (!X || !U || !mu || !K || !d || !tiles_workspace) return fail(DPILQR_EINVAL, "backward_pass: NULL pointer");
    rc = launch_make_tiles(*desc, X, U, tiles_workspace, nullptr, nullptr, desc->B, false, false, as_stream(stream));
    if (rc) return rc;
    return launch_riccati(desc->B, desc->T, desc->k * desc->n_s, desc->k * desc->n_c, tiles_workspace, mu, K, d, nullptr,
                          nullptr, nullptr, desc->B, 0, desc->n_s, desc->n_c, as_stream(stream));
}

int32_t dpilqr_forward_pass(const dpilqr_batch_desc* desc, const double* X, const double* U, const double* K,
                            const double* d, const double* alphas, int32_t n_alpha, double* Xn, double* Un, double* Jn,
                            void* stream) {
    int32_t rc = check_desc(desc);
    if (rc) return rc;
    if (!X || !U || !K || !d || !alphas || !Xn || !Un || !Jn) return fail(DPILQR_EINVAL, "forward_pass: NULL pointer");
    if (n_alpha < 1) return fail(DPILQR_EINVAL, "forward_pass: n_alpha=%d", n_alpha);
    SolveState S{};
    return launch_forward(*desc, kModeCandidates, nullptr, const_cast<double*>(X), const_cast<double*>(U), K, d, alphas,
                          n_alpha, Xn, Un, Jn, S, nullptr, nullptr, desc->B, as_stream(stream));
}

int32_t dpilqr_alphas(double* alphas_host) {
    if (!alphas_host) return fail(DPILQR_EINVAL, "alphas: NULL pointer");
    alpha_table(alphas_host);
    return DPILQR_OK;
}

int64_t dpilqr_solve_workspace_bytes(const dpilqr_batch_desc* desc, int32_t window, int32_t gains_in_workspace) {
    if (!desc || desc->B < 0 || desc->k < 1 || desc->T < 1) return fail(DPILQR_EINVAL, "solve_workspace_bytes: bad desc");
    return (int64_t)SolveWorkspace(*desc, window_of(*desc, window), gains_in_workspace != 0).total;
}

int32_t dpilqr_solve_batch(const dpilqr_batch_desc* desc, const double* x0, double* U, int32_t n_lqr_iter, double tol,
                           int32_t window, void* workspace, int64_t workspace_bytes, double* X, double* J,
                           int32_t* status, int32_t* n_bwd, int32_t* n_fwd, double* trace, double* K_out, double* d_out,
                           void* stream) {
    int32_t rc = check_desc(desc);
    if (rc) return rc;
    if (desc->B == 0) return DPILQR_OK;   // an empty batch: nothing to read or write
    if (!x0 || !U || !X || !J || !status || !n_bwd || !n_fwd || !workspace)
        return fail(DPILQR_EINVAL, "solve_batch: NULL pointer");
    if ((K_out == nullptr) != (d_out == nullptr)) return fail(DPILQR_EINVAL, "solve_batch: K_out and d_out go together");
    if (n_lqr_iter < 0 || n_lqr_iter > kMaxLqrIter) return fail(DPILQR_EINVAL, "solve_batch: n_lqr_iter=%d", n_lqr_iter);
    const dpilqr_batch_desc& D = *desc;
    const int Wn = window_of(D, window);
    const bool gains_by_item = K_out != nullptr;
    const SolveWorkspace W(D, Wn, !gains_by_item);
    if (workspace_bytes < (int64_t)W.total)
        return fail(DPILQR_EWORKSPACE, "solve_batch: workspace %lld B < required %zu B", (long long)workspace_bytes, W.total);
    if (D.B == 0) return DPILQR_OK;
    hipStream_t st = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    double* tiles = reinterpret_cast<double*>(ws + W.tiles);
    double* K = gains_by_item ? K_out : reinterpret_cast<double*>(ws + W.K);
    double* d = gains_by_item ? d_out : reinterpret_cast<double*>(ws + W.d);
    double* alphas = reinterpret_cast<double*>(ws + W.alphas);
    double* Xc = reinterpret_cast<double*>(ws + W.Xc);
    double* Uc = reinterpret_cast<double*>(ws + W.Uc);
    int32_t* lists = reinterpret_cast<int32_t*>(ws + W.lists);
    int32_t* counts = reinterpret_cast<int32_t*>(ws + W.counts);
    int32_t* singular = reinterpret_cast<int32_t*>(ws + W.singular);
    SolveState S{};
    S.mu = reinterpret_cast<double*>(ws + W.mu);
    S.delta = reinterpret_cast<double*>(ws + W.delta);
    S.J_star = reinterpret_cast<double*>(ws + W.J_star);
    S.J_last = reinterpret_cast<double*>(ws + W.J_last);
    S.status = status; S.n_bwd = n_bwd; S.n_fwd = n_fwd; S.trace = trace; S.singular = singular;
    S.n_lqr_iter = n_lqr_iter; S.tol = tol; S.gains_by_item = gains_by_item ? 1 : 0;
    const int n = D.k * D.n_s, m = D.k * D.n_c;

    if (!g_mail.host) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&g_mail.host), sizeof(int32_t) * 2 * kMailRing, hipHostMallocDefault));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&g_mail.dev), g_mail.host, 0));
        for (auto& e : g_mail.ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    g_mail.hist.clear();

    double a[DPILQR_N_ALPHA];
    alpha_table(a);
    const int init_n = D.B > kCountRing + 1 ? D.B : kCountRing + 1;
    hipLaunchKernelGGL(k_init_state, dim3((init_n + 255) / 256), dim3(256), 0, st, D.B, S.mu, S.delta, status, n_bwd, n_fwd,
                       singular, counts, kCountRing + 1, alphas, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]);
    HIP_TRY(hipGetLastError());
    // X, J* <- rollout(x0, U) for every item up front (control.py:164)
    g_prof.begin(3, -1, st);
    rc = launch_forward(D, kModeRollout, x0, X, U, nullptr, nullptr, nullptr, 1, nullptr, nullptr, S.J_star, S, nullptr,
                        nullptr, D.B, st);
    if (rc) return rc;
    g_prof.end(st);
    hipLaunchKernelGGL(k_copy_f64, dim3((D.B + 255) / 256), dim3(256), 0, st, D.B, S.J_star, S.J_last);

    // Iteration loop with continuous admission.  At most Wn sub-problems are in flight; one global
    // iteration = one backward pass + one line search for every active item.  Items that finish are
    // retired by the line-search kernel (it pushes only the survivors onto the next list) and k_admit
    // refills their places from the not-yet-started items, so every launch stays at Wn items although
    // the items need very different numbers of iterations.  The active set lives on the device; the
    // host launches Wn-wide grids (surplus workgroups exit at once) and only reads {active, admitted}
    // one iteration late -- a full iteration of launches is always queued while it waits -- to learn
    // when everything has been started and nothing is left active.
    int32_t* admitted_dev = counts + kCountRing;
    size_t n_iterations = 0;
    if (n_lqr_iter > 0) {
        // the tile producer of the loop writes only structurally non-zero entries: put the zeros in place once
        HIP_TRY(hipMemsetAsync(tiles, 0, W.K - W.tiles, st));
        // One linear model and one R for the whole batch: A, B and L_uu are the same in every record of every
        // item, so they are written once into all Wn slots here (with items 0..Wn-1 as stand-ins; their (X, U)
        // dependent entries are overwritten by each iteration's producer launch) and skipped afterwards.
        static const bool no_static = getenv("DPILQR_TILES_NO_STATIC") != nullptr;   // A/B switch
        const int um = hint_model(D);
        const bool static_part_placed = !no_static && D.R_bstride == 0 &&
                                        (um == kDoubleInt4D || um == kDoubleInt6D || um == kHumanLin6D);
        if (static_part_placed && (rc = launch_make_tiles(D, X, U, tiles, nullptr, nullptr, Wn, true, false, st))) return rc;
        int upper = Wn;
        for (int it = 0; it < kMaxGlobalIter; ++it) {
            n_iterations = (size_t)it + 1;
            int32_t* cur = lists + (size_t)(it & 1) * Wn;
            int32_t* cur_n = counts + (it % kCountRing);
            int32_t* nxt_n = counts + ((it + 1) % kCountRing);
            hipLaunchKernelGGL(k_admit, dim3(1), dim3(256), 0, st, cur, cur_n, nxt_n, admitted_dev, D.B, Wn,
                               g_mail.dev + 2 * (it % kMailRing));
            HIP_TRY(hipEventRecord(g_mail.ev[it % kMailRing], st));
            S.next_items = lists + (size_t)((it + 1) & 1) * Wn;
            S.next_count = nxt_n;
            g_prof.begin(0, it, st);
            if ((rc = launch_make_tiles(D, X, U, tiles, cur, cur_n, upper, true, static_part_placed, st))) return rc;
            g_prof.end(st);
            g_prof.begin(1, it, st);
            if ((rc = launch_riccati(D.B, D.T, n, m, tiles, S.mu, K, d, singular, cur, cur_n, upper, S.gains_by_item, D.n_s, D.n_c,
                                     st)))
                return rc;
            g_prof.end(st, g_sweep_waves);
            g_prof.begin(2, it, st);
            if ((rc = launch_forward(D, kModeLineSearch, nullptr, X, U, K, d, alphas, DPILQR_N_ALPHA, Xc, Uc, nullptr, S,
                                     cur, cur_n, upper, st)))
                return rc;
            g_prof.end(st);
            bool done = false;
            if (it >= kHostLag) {  // {active, admitted} of iteration it - kHostLag
                const int slot = (it - kHostLag) % kMailRing;
                HIP_TRY(hipEventSynchronize(g_mail.ev[slot]));
                const int32_t act = g_mail.host[2 * slot], adm = g_mail.host[2 * slot + 1];
                g_mail.hist.push_back(act);
                // everything started and the list already empty back then: the iterations since were no-ops
                done = (adm >= D.B && act == 0);
                // once everything is admitted the list can only shrink: tighten the grid
                upper = (adm >= D.B) ? std::min(Wn, std::max(act, 1)) : Wn;
            }
            if (done) break;
            if (it + 1 == kMaxGlobalIter) return fail(DPILQR_EUNSUPPORTED, "solve_batch: more than %d global iterations", kMaxGlobalIter);
        }
    }
    hipLaunchKernelGGL(k_copy_f64, dim3((D.B + 255) / 256), dim3(256), 0, st, D.B, S.J_last, J);
    if (n_lqr_iter == 0) hipLaunchKernelGGL(k_finish_status, dim3((D.B + 255) / 256), dim3(256), 0, st, D.B, status);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    if (n_lqr_iter > 0) {
        // the last iterations' exact lengths were posted but not yet consumed
        for (size_t j = g_mail.hist.size(); j < n_iterations; ++j) g_mail.hist.push_back(g_mail.host[2 * (j % kMailRing)]);
    }
    g_prof.collect(g_mail.hist, D.B);
    return DPILQR_OK;
}

int32_t dpilqr_debug_stamps(void* buf) {
    void* p = buf;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)));
    return DPILQR_OK;
}

int32_t dpilqr_profile_enable(int32_t enable) {
    const int32_t prev = g_prof.on ? 1 : 0;
    g_prof.on = (enable & 1) != 0;
    g_prof.mask = (enable >> 1) & 0xF ? (enable >> 1) & 0xF : 0xF;
    return prev;
}

int32_t dpilqr_profile_read(double ms[4], int64_t launches[4], int64_t items[4], int32_t reset) {
    if (!ms || !launches || !items) return fail(DPILQR_EINVAL, "profile_read: NULL pointer");
    for (int c = 0; c < 4; ++c) {
        ms[c] = g_prof.ms[c]; launches[c] = g_prof.launches[c]; items[c] = g_prof.items[c];
        if (reset) { g_prof.ms[c] = 0; g_prof.launches[c] = 0; g_prof.items[c] = 0; }
    }
    return DPILQR_OK;
}

int32_t dpilqr_profile_read_sweep(int32_t waves, double* ms, int64_t* launches, int64_t* items, int32_t reset) {
    if (!ms || !launches || !items) return fail(DPILQR_EINVAL, "profile_read_sweep: NULL pointer");
    if (waves != 4 && waves != 8 && waves != 12) return fail(DPILQR_EINVAL, "profile_read_sweep: waves=%d (4, 8 or 12)", waves);
    const int v = waves / 4 - 1;
    *ms = g_prof.sweep_ms[v]; *launches = g_prof.sweep_launches[v]; *items = g_prof.sweep_items[v];
    if (reset) { g_prof.sweep_ms[v] = 0; g_prof.sweep_launches[v] = 0; g_prof.sweep_items[v] = 0; }
    return DPILQR_OK;
}

int32_t dpilqr_pairwise_graph(int32_t S, int32_t N, int32_t k, int32_t n_s, const double* X, const double* radius,
                              int32_t* adj, void* stream) {
    if (S < 0 || N < 1 || k < 1 || n_s < 2 || !X || !radius || !adj) return fail(DPILQR_EINVAL, "pairwise_graph: bad argument");
    if (S == 0) return DPILQR_OK;
    hipStream_t st = as_stream(stream);
    HIP_TRY(hipMemsetAsync(adj, 0, sizeof(int32_t) * (size_t)S * k * k, st));
    const int64_t total = (int64_t)S * (k * (k - 1) / 2 + k);
    hipLaunchKernelGGL(k_pairwise_graph, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, st, S, N, k, n_s, X, radius, adj);
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

}  // extern "C"
