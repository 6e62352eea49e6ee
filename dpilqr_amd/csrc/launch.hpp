// launch.hpp -- host-side declarations shared by the translation units of libdpilqr_hip.so.
//
// The library is built from several .hip files so that they compile in parallel (one file with every kernel
// instantiation took over two minutes): tu_tiles.hip (K1), tu_riccati.hip (K2), tu_forward.hip (K3, rollouts, the small
// batched entry points), tu_big.hip (the sweep for n_x > 60 and the fp32 arm), tu_team.hip (the fused
// wavefront sweeps with a helper wavefront per item) and dpilqr_hip.hip (the C ABI and the
// solve loop); round 4 added tu_inprod.hip (the wavefront sweeps with in-sweep production) and tu_lsteam.hip (the line search
// with two wavefronts per item).  No device code crosses a file boundary.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "dpilqr_hip.h"
#include "solve_state.hpp"

namespace dpilqr {

int32_t fail(int32_t code, const char* fmt, ...);   // records the calling thread's error message, returns `code`
const char* last_error();

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return ::dpilqr::fail(DPILQR_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int kMaxLds = 160 * 1024;  // gfx950: 160 KiB per workgroup

template <typename Kern>
inline int32_t allow_lds(Kern kern, size_t bytes) {
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
    default: return ::dpilqr::fail(DPILQR_EINVAL, "unsupported per-agent state dim %d", (int)(ns)); \
    }

// compute units of the current device (256 on MI355X): the sweep deals its items over rounds of this many workgroups
int device_cus();

// A/B route switches and tuning knobs (DPILQR_NO_FUSED, DPILQR_MFMA_WAVES, DPILQR_BIG_TEAM, ...) exist for experiments and
// tests.  They are honoured only in a process that sets DPILQR_DEBUG_ROUTES=1: without that gate route_env answers null for
// every name, so a stray variable in a production process's environment cannot change which kernel serves a batch.  The gate and
// the library's only getenv live in dpilqr_hip.hip.
const char* route_env(const char* name);
inline bool route_flag(const char* name) { return route_env(name) != nullptr; }
inline int route_int(const char* name, int dflt) { const char* e = route_env(name); return (e && *e) ? atoi(e) : dflt; }

// hints packed into dpilqr_batch_desc::uniform_model (include/dpilqr_hip.h): -1 = unknown / mixed
inline int hint_model(const dpilqr_batch_desc& D) { return (D.uniform_model & 0xff) - 1; }
inline int hint_n_dims(const dpilqr_batch_desc& D) { return ((D.uniform_model >> 8) & 0xff) - 1; }
inline bool hint_shared_weights(const dpilqr_batch_desc& D) { return ((D.uniform_model >> 16) & 1) != 0; }
inline bool hint_planar4(const dpilqr_batch_desc& D) { return ((D.uniform_model >> 17) & 1) != 0 || hint_model(D) == 0 || hint_model(D) == 3; }
// The fused sweeps (no tile records: linearize / quadraticize evaluated inside the sweep).
// Wavefront sweep (n_x <= 20, riccati_mfma.hpp): DoubleIntDynamics4D agents only, planar proximity cost, one Q / R / Q_f for
// every agent of every item (the descriptor's hints), at most five agents.
inline bool fused_wavefront_sweep_applies(const dpilqr_batch_desc& D) {
    static const bool off = route_flag("DPILQR_NO_FUSED");   // A/B switch: the record-fed sweep
    return !off && hint_model(D) == 0 && hint_n_dims(D) == 2 && hint_shared_weights(D) && D.Q_bstride == 0 && D.R_bstride == 0 &&
           D.Qf_bstride == 0 && D.n_s == 4 && D.n_c == 2 && D.k <= 5;
}
// ... and its general form (k_riccati_mfma_general, FUSED == 2): at most five agents of the planar four-state models --
// DoubleIntDynamics4D and UnicycleDynamics4D, mixed or not (the descriptor's hints) -- with any per-agent, per-item Q / R / Q_f,
// planar proximity cost.  Two wavefronts per SIMD at most (the per-agent weights take the LDS the third one needs).
inline bool fused_wavefront_general_applies(const dpilqr_batch_desc& D) {
    static const bool off = route_flag("DPILQR_NO_FUSED") || route_flag("DPILQR_NO_FUSED_GENERAL");
    return !off && hint_planar4(D) && hint_n_dims(D) == 2 && D.n_s == 4 && D.n_c == 2 && D.k <= 5;
}
// Workgroup sweep (riccati_wg.hpp): 6..15 agents of the four-state family or 2..10 of the six-state family -- any models of
// the family, any per-agent weights, any n_dims.
inline bool fused_workgroup_sweep_applies(const dpilqr_batch_desc& D) {
    static const bool off = route_flag("DPILQR_NO_FUSED") || route_flag("DPILQR_NO_FUSED_WG");
    return !off && ((D.n_s == 4 && D.n_c == 2 && D.k >= 6 && D.k <= 15) || (D.n_s == 6 && D.n_c == 3 && D.k >= 2 && D.k <= 10));
}
// Wavefront sweep with in-sweep production (riccati_mfma.hpp, PNS; tu_inprod.hip): at most four agents of the six-state family
// (n_x <= 24) or at most six CarDynamics3D agents (n_x <= 18) -- any models of the family, any per-agent weights, any n_dims --
// padded into the next instantiated size.
inline bool fused_wavefront_inprod_applies(const dpilqr_batch_desc& D) {
    static const bool off = route_flag("DPILQR_NO_FUSED") || route_flag("DPILQR_NO_INPROD");
    static const bool no4 = route_flag("DPILQR_NO_INPROD4");   // A/B switch: the four-state clusters' previous routes
    if (off) return false;
    // ... and what is left of the four-state family at n_x <= 20: at most five agents WITHOUT the hints the forms above need (a
    // proximity cost over three dimensions, mixed models unannounced).  (Six agents, n_x = 24, keep the producer + the
    // record-fed wavefront sweep: a full 2048-item pass is 2.18 ms in-sweep against 1.94 + 1.22, but a whole solve 81.5 against
    // 72.6 ms -- fifteen pairs' derivatives at the top of each of T = 100 steps of a lone wavefront in the solve's long tail;
    // profiles/r04_inprod_four_state.txt)
    if (D.n_s == 4 && D.n_c == 2 && D.k <= 5 && !no4)
        return !fused_wavefront_sweep_applies(D) && !fused_wavefront_general_applies(D);
    return (D.n_s == 6 && D.n_c == 3 && D.k <= 4) || (D.n_s == 3 && D.n_c == 2 && D.k <= 6);
}
inline bool fused_sweep_applies(const dpilqr_batch_desc& D) {
    return fused_wavefront_sweep_applies(D) || fused_wavefront_general_applies(D) || fused_workgroup_sweep_applies(D) ||
           fused_wavefront_inprod_applies(D);
}
// The solve loop's choice where both a record-free workgroup sweep and a record-fed WAVEFRONT sweep serve a batch: at
// n_x = 12 and 24 (two / four six-state agents: cfg4's small clusters; six four-state agents: cfg3's smallest) a wavefront
// per item beats a workgroup per item by more than the tile producer costs.  2048 items, one sweep (profiles/
// r03_small_clusters.txt): n_x = 12 fused workgroup sweep 1.74 ms against producer 0.36 + wavefront sweep 0.26 ms; n_x = 24
// (four quadcopters) 2.67 against 1.01 + 0.88.  Whole solves of 2048 items: two quadcopters 24.3 -> 13.4 ms, four 26.7 -> 20.6,
// six unicycles 86.2 -> 80.6, six double integrators 23.8 -> 21.3.  DPILQR_NO_WAVE_PREF: A/B switch.
inline bool solve_prefers_records(const dpilqr_batch_desc& D) {
    static const bool off = route_flag("DPILQR_NO_WAVE_PREF");
    // round 4: three six-state agents (n_x = 18: 16 % of cfg4's sub-problems) through the (20, 10) wavefront sweep, padded
    // while loading (riccati_mfma.hpp, PAD; DPILQR_RICCATI_NO_PAD switches it off in the launcher)
    // (two .. four six-state agents: records unless the in-sweep producer serves them)
    if (fused_wavefront_inprod_applies(D)) { static const bool rec = route_flag("DPILQR_INPROD_RECORDS"); return rec; }
    return !off && ((D.n_s == 6 && D.n_c == 3 && (D.k == 2 || D.k == 3 || D.k == 4)) || (D.n_s == 4 && D.n_c == 2 && D.k == 6));
}

// ---- tu_tiles.hip
int32_t launch_make_tiles(const dpilqr_batch_desc& D, const double* X, const double* U, double* tiles,
                          const int32_t* items, const int32_t* n_items, int grid_items, bool sparse, bool dyn_only,
                          hipStream_t st);
int32_t tile_layout_host(int32_t n_x, int32_t n_u, int64_t offsets[7], int64_t row_strides[7], int64_t* stride);

// ---- tu_riccati.hip
extern thread_local int g_sweep_waves;   // wavefronts per workgroup of the last launch_riccati (0: not the wavefront sweep)
int32_t launch_riccati(int B, int T, int n, int m, const double* tiles, const double* mu, double* K, double* d,
                       int32_t* singular, const int32_t* items, const int32_t* n_items, int grid_items,
                       int gains_by_item, int block_ns, int block_nc, hipStream_t st);
int32_t launch_riccati_fused(const dpilqr_batch_desc& D, const double* X, const double* U, const double* mu, double* K,
                             double* d, int32_t* singular, const int32_t* items, const int32_t* n_items, int grid_items,
                             int gains_by_item, hipStream_t st);
int32_t set_stamp_buffer_riccati(void* device_buffer);
// ---- tu_team.hip: the fused wavefront sweeps with a helper wavefront per item, for launches of at most 1024 items
int32_t launch_riccati_team(const dpilqr_batch_desc& D, const double* X, const double* U, const double* mu, double* K, double* d,
                            int32_t* singular, const int32_t* items, const int32_t* n_items, int grid_items, int gains_by_item,
                            hipStream_t st);

// ---- tu_inprod.hip: the wavefront sweeps with in-sweep production (six-state family up to four agents, CarDynamics3D up to six)
int32_t launch_riccati_inprod(const dpilqr_batch_desc& D, const double* X, const double* U, const double* mu, double* K, double* d,
                              int32_t* singular, const int32_t* items, const int32_t* n_items, int grid_items, int gains_by_item,
                              hipStream_t st);

// ---- tu_lsteam.hip: the line search with two wavefronts per item (rollout / costs), for launches of at most 1024 items
int32_t launch_linesearch_team(const dpilqr_batch_desc& D, double* X, double* U, const double* K, const double* d,
                               const double* alphas, double* Xc, double* Uc, const SolveState& S, const int32_t* items,
                               const int32_t* n_items, int grid_items, hipStream_t st);

// ---- tu_forward.hip
int32_t launch_forward(const dpilqr_batch_desc& D, int mode, const double* x0, double* X, double* U, const double* K,
                       const double* d, const double* alphas, int ngrp, double* Xc, double* Uc, double* Jc,
                       const SolveState& S, const int32_t* items, const int32_t* n_items, int grid_items,
                       hipStream_t st);
int32_t set_stamp_buffer_forward(void* device_buffer);
int32_t launch_model_op(int op, int32_t n, int32_t ns, const int32_t* model, const double* x, const double* u, double dt,
                        double* o1, double* o2, hipStream_t st);
int32_t launch_cost_eval(const dpilqr_batch_desc& D, int32_t n_pts, const double* x, const double* u, int32_t terminal,
                         double* cost, hipStream_t st);
int32_t launch_pairwise_graph(int32_t S, int32_t N, int32_t k, int32_t n_s, const double* X, const double* radius,
                              int32_t* adj, hipStream_t st);


// ---- tu_big.hip: large clusters (n_x > 60) in fp64, any size in fp32 (BASELINE config 5's tolerance study)
int64_t riccati_big_scratch_elems(int n, int m);   // per sub-problem in flight, in elements of the arithmetic type
int32_t launch_riccati_big_f64(const dpilqr_batch_desc& D, const double* X, const double* U, const double* mu, double* K,
                               double* d, int32_t* singular, const int32_t* items, const int32_t* n_items,
                               int grid_items, int gains_by_item, void* scratch, hipStream_t st);
int32_t launch_riccati_big_f32(const dpilqr_batch_desc& D, const float* X, const float* U, const double* mu, float* K,
                               float* d, int32_t* singular, const int32_t* items, const int32_t* n_items,
                               int grid_items, int gains_by_item, void* scratch, hipStream_t st);
int32_t launch_forward_big_f64(const dpilqr_batch_desc& D, int mode, const double* x0, double* X, double* U, const double* K,
                               const double* d, const double* alphas, int ngrp, double* Xc, double* Uc, double* Jc,
                               const SolveState& S, const int32_t* items, const int32_t* n_items, int grid_items,
                               hipStream_t st);
int32_t launch_forward_big_f32(const dpilqr_batch_desc& D, int mode, const float* x0, float* X, float* U, const float* K,
                               const float* d, const double* alphas, int ngrp, float* Xc, float* Uc, double* Jc,
                               const SolveState& S, const int32_t* items, const int32_t* n_items, int grid_items,
                               hipStream_t st);
// the sweep's choice: clusters beyond the wavefront / workgroup sweeps (n_x > 60) take the fused big kernel
inline bool uses_big_path(int n_x) {
    static const bool force = route_flag("DPILQR_FORCE_BIG");   // test / A-B switch: every size through tu_big.hip
    return force || n_x > 60;
}

}  // namespace dpilqr
