// tu_forward.hip -- K3: rollouts, forward passes and the line search (forward.hpp, forward_wave.hpp), the small
// batched entry points (model FFI, cost evaluation, interaction graph), and their launchers.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "launch.hpp"
#include "forward.hpp"
#include "forward_wave.hpp"

namespace dpilqr {

static int forward_threads(int k, int ngrp) {
    const int t = ((k * ngrp + 63) / 64) * 64;
    return t;
}

static bool no_wave_ro() { static const bool v = route_flag("DPILQR_FORWARD_GENERIC"); return v; }

int32_t launch_forward(const dpilqr_batch_desc& D, int mode, const double* x0, double* X, double* U, const double* K,
                       const double* d, const double* alphas, int ngrp, double* Xc, double* Uc, double* Jc,
                       const SolveState& S, const int32_t* items, const int32_t* n_items, int grid_items,
                       hipStream_t st) {
    if (grid_items <= 0) return DPILQR_OK;
    const int n = D.k * D.n_s, m = D.k * D.n_c;
    int threads = forward_threads(D.k, mode == kModeRollout ? 1 : ngrp);
    if (threads > 256) return fail(DPILQR_EUNSUPPORTED, "k*n_alpha=%d exceeds the 256-thread workgroup of the forward pass", D.k * ngrp);
    const size_t lds_item = (forward_lds_bytes(n, m, D.k, ngrp) + 15) & ~(size_t)15;
    // large clusters: K[t] does not fit the per-step LDS staging -- the variant that reads it from global memory (tu_big.hip)
    if (uses_big_path(n) || lds_item > (size_t)kMaxLds || (mode != kModeRollout && (m * n + threads - 1) / threads > kMaxStage))
        return launch_forward_big_f64(D, mode, x0, X, U, K, d, alphas, ngrp, Xc, Uc, Jc, S, items, n_items, grid_items, st);
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
    static const bool no_wave = route_flag("DPILQR_FORWARD_GENERIC");   // A/B switch
    if (!no_wave && mode == kModeLineSearch && hint_model(D) >= 0 && ngrp == DPILQR_N_ALPHA && items && n_items) {
        const int model = hint_model(D);
        {   // launches of at most one item per SIMD: two wavefronts per item, rollout and costs (tu_lsteam.hip)
            const int32_t rc_t = launch_linesearch_team(D, X, U, K, d, alphas, Xc, Uc, S, items, n_items, grid_items, st);
            if (rc_t != DPILQR_EUNSUPPORTED) return rc_t;
        }
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
    static const bool no_pack = route_flag("DPILQR_FORWARD_NO_PACK");   // diagnostic switch
    const int ipb = (!no_pack && threads == 64 && 4 * lds_item <= (size_t)kMaxLds) ? 4 : 1;
    const size_t lds = lds_item * ipb;
    threads *= ipb;
    DISPATCH_FAMILY(D.n_s, {
        int32_t rc = allow_lds(k_forward<double, NS, NC, false>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((k_forward<double, NS, NC, false>), dim3((grid_items + ipb - 1) / ipb), dim3(threads), lds, st, D, mode,
                           x0, X, U, K, d, alphas, ngrp, Xc, Uc, Jc, S, items, n_items, ipb, (int)(lds_item / sizeof(double)));
    })
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t set_stamp_buffer_forward(void* buf) {
    void* p = buf;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)));
    return DPILQR_OK;
}

int32_t launch_model_op(int op, int32_t n, int32_t ns, const int32_t* model, const double* x, const double* u, double dt,
                        double* o1, double* o2, hipStream_t st) {
    const dim3 grid((n + 63) / 64), block(64);
    DISPATCH_FAMILY(ns, {
        if (op == 0) hipLaunchKernelGGL((k_model_op<NS, NC, 0>), grid, block, 0, st, n, model, x, u, dt, o1, o2);
        else if (op == 1) hipLaunchKernelGGL((k_model_op<NS, NC, 1>), grid, block, 0, st, n, model, x, u, dt, o1, o2);
        else hipLaunchKernelGGL((k_model_op<NS, NC, 2>), grid, block, 0, st, n, model, x, u, dt, o1, o2);
    })
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t launch_cost_eval(const dpilqr_batch_desc& D, int32_t n_pts, const double* x, const double* u, int32_t terminal,
                         double* cost, hipStream_t st) {
    const int64_t total = (int64_t)D.B * n_pts;
    const dim3 grid((unsigned)((total + 63) / 64)), block(64);
    DISPATCH_FAMILY(D.n_s, {
        hipLaunchKernelGGL((k_cost_eval<NS, NC>), grid, block, 0, st, D, n_pts, x, u, terminal, cost);
    })
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t launch_pairwise_graph(int32_t S, int32_t N, int32_t k, int32_t n_s, const double* X, const double* radius,
                              int32_t* adj, hipStream_t st) {
    HIP_TRY(hipMemsetAsync(adj, 0, sizeof(int32_t) * (size_t)S * k * k, st));
    const int64_t total = (int64_t)S * (k * (k - 1) / 2 + k);
    hipLaunchKernelGGL(k_pairwise_graph, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, st, S, N, k, n_s, X, radius, adj);
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

}  // namespace dpilqr
