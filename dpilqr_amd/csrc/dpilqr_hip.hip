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
#include <new>
#include <vector>

#include "dpilqr_hip.h"
#include "launch.hpp"
#include "models.hpp"    // model_ns / model_nc (host-visible)
#include "tiles.hpp"     // TileLayout (host-visible)

namespace dpilqr {

static thread_local char g_err[512] = "";

int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
const char* last_error() { return g_err; }

// The one gate in front of every A/B route switch (launch.hpp): closed unless DPILQR_DEBUG_ROUTES=1, read once.
const char* route_env(const char* name) {
    static const bool open = [] { const char* g = std::getenv("DPILQR_DEBUG_ROUTES"); return g && g[0] == '1' && g[1] == 0; }();
    return open ? std::getenv(name) : nullptr;
}

// compute units of the current device (256 on MI355X), remembered per device
int device_cus() {
    static int cus_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus_of[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus_of[dev] = n;
    }
    return cus_of[dev];
}

}  // namespace dpilqr

using namespace dpilqr;

namespace {

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
// mail[2] = the finished PREFIX: items are admitted in index order, so every item below the smallest index still on the
// list has finished and its results are final in memory (the progress callback of dpilqr_solver_set_progress).
// want_prefix = 0: nobody listens for progress, the scan of the list for its lowest index is skipped (mail[2] = 0).
// t_admit (null without t_kill): the admitted items' own t0 of control.py:167 -- the device's constant-rate clock now,
// just before their first backward pass.
// mail[3] = items retired with DPILQR_STATUS_FAULT so far (solve_state.hpp: retire_without_gains).
__global__ void k_admit(int32_t* list, int32_t* count, int32_t* next_count, int32_t* admitted, int B, int window,
                        int32_t* mail, int want_prefix, int64_t* t_admit, const int32_t* fault) {
    const int base = *count, first = *admitted;
    const int n_new = min(B - first, window - base);
    __shared__ int lowest;
    if (threadIdx.x == 0) lowest = first;
    __syncthreads();
    for (int i = threadIdx.x; i < n_new; i += blockDim.x) list[base + i] = first + i;
    if (t_admit && n_new > 0) {
        const int64_t now = (int64_t)__builtin_amdgcn_s_memrealtime();
        for (int i = threadIdx.x; i < n_new; i += blockDim.x) t_admit[first + i] = now;
    }
    if (mail && want_prefix) {
        int lo = first;
        for (int i = threadIdx.x; i < base; i += blockDim.x) lo = min(lo, list[i]);
        for (int off = 32; off > 0; off >>= 1) lo = min(lo, __shfl_down(lo, off));
        if ((threadIdx.x & 63) == 0) atomicMin(&lowest, lo);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *count = base + n_new; *admitted = first + n_new; *next_count = 0;
        if (mail) { mail[0] = base + n_new; mail[1] = first + n_new; mail[2] = want_prefix ? lowest : 0; mail[3] = *fault; }   // pinned host memory: the host reads it after the event
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

// Shape of a solve: which kernels serve it and how large an element is.
struct SolveShape {
    bool big;        // the fused workgroup-per-item sweep of tu_big.hip (n_x > 60, or the fp32 arm): no tile records
    size_t elem;     // bytes per element of the trajectories / gains (8: fp64, 4: fp32)
    bool fused;      // a fused sweep of tu_riccati.hip serves the batch: no tile records, no producer, no workspace for them
};

struct SolveWorkspace {
    // W = window = most sub-problems in flight at once: the big per-iteration buffers (tile records or sweep scratch,
    // gains, line-search candidates) are indexed by position in the active list and sized by W, not by B.
    size_t tiles, K, d, Xc, Uc, mu, delta, J_star, J_last, alphas, singular, lists, counts, t_admit, total;
    SolveWorkspace(const dpilqr_batch_desc& D, int W, bool gains_in_ws, SolveShape sh) {
        const size_t B = D.B, n = (size_t)D.k * D.n_s, m = (size_t)D.k * D.n_c, T = D.T, Wn = W, e = sh.elem;
        const TileLayout L((int)n, (int)m);
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        size_t o = 0;
        tiles = o;    o = al(o + (sh.big ? e * Wn * (size_t)riccati_big_scratch_elems((int)n, (int)m)
                                         : (sh.fused ? 0 : sizeof(double) * Wn * (T + 1) * L.stride)));
        K = o;        o = al(o + (gains_in_ws ? e * Wn * T * m * n : 0));
        d = o;        o = al(o + (gains_in_ws ? e * Wn * T * m : 0));
        // line-search candidates: every alpha's trajectory, so that accepting one is a copy, not a re-roll
        Xc = o;       o = al(o + e * Wn * DPILQR_N_ALPHA * (T + 1) * n);
        Uc = o;       o = al(o + e * Wn * DPILQR_N_ALPHA * T * m);
        mu = o;       o = al(o + sizeof(double) * B);
        delta = o;    o = al(o + sizeof(double) * B);
        J_star = o;   o = al(o + sizeof(double) * B);
        J_last = o;   o = al(o + sizeof(double) * B);
        alphas = o;   o = al(o + sizeof(double) * DPILQR_N_ALPHA);
        singular = o; o = al(o + sizeof(int32_t) * B);
        lists = o;    o = al(o + sizeof(int32_t) * 2 * Wn);
        counts = o;   o = al(o + sizeof(int32_t) * (kCountRing + 2));   // ring + the `admitted` counter + the fault counter
        t_admit = o;  o = al(o + sizeof(int64_t) * B);                  // t_kill: every item's admission stamp
        total = o;
    }
};
constexpr int kMaxLqrIter = 4096;
constexpr int kMaxGlobalIter = 1 << 20;  // launches of the iteration loop one solve_batch call may make

int window_of(const dpilqr_batch_desc& D, int window) { return (window <= 0 || window > D.B) ? (D.B > 0 ? D.B : 1) : window; }

// The host reads the device-side counters kHostLag iterations late: that many iterations of launches are always
// queued behind the one the GPU is running, so a host thread that loses its core for a millisecond (a loaded box)
// does not leave the GPU idle.  The price is kHostLag empty iterations (a dozen tiny launches) at the end of a solve.
constexpr int kHostLag = 3, kMailRing = kHostLag + 1, kMailWords = 4;   // {active, admitted, finished prefix, faulted items}

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
    ~Profiler() {
        for (auto& e : pool) (void)hipEventDestroy(e);
    }
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
    void drop() { recs.clear(); used = 0; }   // a failed solve: forget the half-recorded pairs
    // after the stream has been synchronised; active[it] = items processed by iteration it
    void collect(const std::vector<int32_t>& active, int B) {
        if (!on) { drop(); return; }
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
        drop();
    }
};

}  // namespace

// Host-side state of the synchronous solve (dpilqr_solver_create): the pinned mailbox the admission kernel posts the
// active-list counters into, the events that tell the host when a post has landed, and the optional profiler.  Bound
// to the device that was current when it was created.
struct dpilqr_solver {
    int device = -1;
    int32_t* host = nullptr;
    int32_t* dev = nullptr;      // the same words as the device sees them
    std::vector<int32_t> hist;   // exact active-list length of every global iteration of the last solve
    hipEvent_t ev[kMailRing] = {};
    Profiler prof;
    dpilqr_progress_fn progress = nullptr;   // dpilqr_solver_set_progress
    void* progress_user = nullptr;
    int32_t init() {
        HIP_TRY(hipGetDevice(&device));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&host), sizeof(int32_t) * kMailWords * kMailRing, hipHostMallocDefault));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0));
        for (auto& e : ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return DPILQR_OK;
    }
    void release() {
        if (host) (void)hipHostFree(host);
        host = nullptr;
        for (auto& e : ev) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
    }
    ~dpilqr_solver() { release(); }
};

namespace {

// dpilqr_solve_batch(solver = NULL): one default solver per host thread, created on first use, re-created when the
// thread has moved to another device
thread_local dpilqr_solver g_default_solver;

int32_t default_solver(dpilqr_solver** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (g_default_solver.host && g_default_solver.device != dev) g_default_solver.release();
    if (!g_default_solver.host) {
        int32_t rc = g_default_solver.init();
        if (rc) return rc;
    }
    *out = &g_default_solver;
    return DPILQR_OK;
}

// At the end of an enqueue-only call: bring the double-buffered active list and the counter ring back to the state a
// call starting at iteration 0 expects (survivors in list 0, their count in counts[0], the other counters zero).
__global__ void k_normalise_lists(int32_t* lists, int32_t* counts, int window, int it_end) {
    const int src = it_end & 1, cs = it_end % kCountRing;
    __shared__ int n_sh;
    if (threadIdx.x == 0) n_sh = counts[cs];
    __syncthreads();
    const int n = n_sh;
    if (src == 1)
        for (int i = threadIdx.x; i < n; i += blockDim.x) lists[i] = lists[window + i];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int c = 0; c < kCountRing; ++c) counts[c] = 0;
        counts[0] = n;
    }
}

// per arithmetic type: the launchers of a solve
template <typename R> struct Passes;
template <> struct Passes<double> {
    static int32_t forward(const dpilqr_batch_desc& D, int mode, const double* x0, double* X, double* U, const double* K,
                           const double* d, const double* alphas, int ngrp, double* Xc, double* Uc, double* Jc,
                           const SolveState& S, const int32_t* items, const int32_t* n_items, int grid, hipStream_t st) {
        return launch_forward(D, mode, x0, X, U, K, d, alphas, ngrp, Xc, Uc, Jc, S, items, n_items, grid, st);
    }
    static int32_t sweep_big(const dpilqr_batch_desc& D, const double* X, const double* U, const double* mu, double* K, double* d,
                             int32_t* sing, const int32_t* items, const int32_t* n_items, int grid, int by_item, void* scratch,
                             hipStream_t st) {
        return launch_riccati_big_f64(D, X, U, mu, K, d, sing, items, n_items, grid, by_item, scratch, st);
    }
};
template <> struct Passes<float> {
    static int32_t forward(const dpilqr_batch_desc& D, int mode, const float* x0, float* X, float* U, const float* K,
                           const float* d, const double* alphas, int ngrp, float* Xc, float* Uc, double* Jc,
                           const SolveState& S, const int32_t* items, const int32_t* n_items, int grid, hipStream_t st) {
        return launch_forward_big_f32(D, mode, x0, X, U, K, d, alphas, ngrp, Xc, Uc, Jc, S, items, n_items, grid, st);
    }
    static int32_t sweep_big(const dpilqr_batch_desc& D, const float* X, const float* U, const double* mu, float* K, float* d,
                             int32_t* sing, const int32_t* items, const int32_t* n_items, int grid, int by_item, void* scratch,
                             hipStream_t st) {
        return launch_riccati_big_f32(D, X, U, mu, K, d, sing, items, n_items, grid, by_item, scratch, st);
    }
};

template <typename R>
SolveShape shape_of(const dpilqr_batch_desc& D) {
    const bool big = sizeof(R) == 4 || uses_big_path(D.k * D.n_s);
    return SolveShape{big, sizeof(R), !big && fused_sweep_applies(D) && !solve_prefers_records(D)};
}

// ilqrSolver.solve for every item of the batch.  solver != NULL: the synchronous, adaptive form (the host follows the
// device's counters a few iterations late and stops launching when everything has finished).  solver == NULL: the
// enqueue-only form -- exactly n_global_iter iterations of launches, no host read, no synchronisation.
template <typename R>
int32_t solve_impl(dpilqr_solver* solver, const dpilqr_batch_desc* desc, const R* x0, R* U, int32_t n_lqr_iter, double tol,
                   double t_kill, int32_t window, void* workspace, int64_t workspace_bytes, R* X, double* J, int32_t* status,
                   int32_t* n_bwd, int32_t* n_fwd, double* trace, R* K_out, R* d_out, int32_t n_global_iter,
                   int32_t resume, hipStream_t st) {
    int32_t rc = check_desc(desc);
    if (rc) return rc;
    if (desc->B == 0) return DPILQR_OK;   // an empty batch: nothing to read or write
    if (!x0 || !U || !X || !J || !status || !n_bwd || !n_fwd || !workspace)
        return fail(DPILQR_EINVAL, "solve_batch: NULL pointer");
    if ((K_out == nullptr) != (d_out == nullptr)) return fail(DPILQR_EINVAL, "solve_batch: K_out and d_out go together");
    if (n_lqr_iter < 0 || n_lqr_iter > kMaxLqrIter) return fail(DPILQR_EINVAL, "solve_batch: n_lqr_iter=%d", n_lqr_iter);
    if (!solver && n_global_iter < 0) return fail(DPILQR_EINVAL, "solve_enqueue: n_global_iter=%d", n_global_iter);
    const dpilqr_batch_desc& D = *desc;
    const int Wn = window_of(D, window);
    const bool gains_by_item = K_out != nullptr;
    const SolveShape sh = shape_of<R>(D);
    const SolveWorkspace W(D, Wn, !gains_by_item, sh);
    if (workspace_bytes < (int64_t)W.total)
        return fail(DPILQR_EWORKSPACE, "solve_batch: workspace %lld B < required %zu B", (long long)workspace_bytes, W.total);
    if (solver) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (dev != solver->device)
            return fail(DPILQR_EINVAL, "solve_batch: the solver belongs to device %d, the current device is %d", solver->device, dev);
    }
    char* ws = static_cast<char*>(workspace);
    double* tiles = reinterpret_cast<double*>(ws + W.tiles);       // tile records, or the big sweep's scratch
    R* K = gains_by_item ? K_out : reinterpret_cast<R*>(ws + W.K);
    R* d = gains_by_item ? d_out : reinterpret_cast<R*>(ws + W.d);
    double* alphas = reinterpret_cast<double*>(ws + W.alphas);
    R* Xc = reinterpret_cast<R*>(ws + W.Xc);
    R* Uc = reinterpret_cast<R*>(ws + W.Uc);
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
    // t_kill in ticks of the constant-rate clock the kernels read (s_memrealtime; the runtime reports its rate in kHz)
    int64_t* t_admit = nullptr;
    S.t_admit = nullptr; S.t_kill_ticks = 0;
    if (t_kill > 0.0) {
        int dev_now = 0, khz = 0;
        HIP_TRY(hipGetDevice(&dev_now));
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev_now) != hipSuccess || khz <= 0) khz = 100000;
        const double ticks = t_kill * 1e3 * (double)khz;
        S.t_kill_ticks = ticks < 1.0 ? 1 : (ticks > 9e18 ? (int64_t)9e18 : (int64_t)ticks);
        t_admit = reinterpret_cast<int64_t*>(ws + W.t_admit);
        S.t_admit = t_admit;
    }
    const int n = D.k * D.n_s, m = D.k * D.n_c;
    Profiler none;
    Profiler& prof = solver ? solver->prof : none;
    if (solver) solver->hist.clear();
    // every failure after the first launch leaves kernels queued that still use the workspace: wait for them before
    // the caller gets its buffers back, and forget the profiler's half-recorded event pairs
    auto bail = [&](int32_t code) {
        (void)hipStreamSynchronize(st);
        prof.drop();
        return code;
    };

    int32_t* admitted_dev = counts + kCountRing;
    S.fault = counts + kCountRing + 1;
    static const bool no_static = route_flag("DPILQR_TILES_NO_STATIC");   // A/B switch
    const int um = hint_model(D);
    // One linear model and one R for the whole batch: A, B and L_uu are the same in every record of every item
    // the fused sweep evaluates the plugins itself: no tile producer, no records
    const bool fused = sh.fused;
    const bool static_part = !fused && !sh.big && !no_static && D.R_bstride == 0 &&
                             (um == kDoubleInt4D || um == kDoubleInt6D || um == kHumanLin6D);
    if (!resume) {
        double a[DPILQR_N_ALPHA];
        alpha_table(a);
        const int init_n = D.B > kCountRing + 2 ? D.B : kCountRing + 2;
        hipLaunchKernelGGL(k_init_state, dim3((init_n + 255) / 256), dim3(256), 0, st, D.B, S.mu, S.delta, status, n_bwd, n_fwd,
                           singular, counts, kCountRing + 2, alphas, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]);
        HIP_TRY(hipGetLastError());
        // X, J* <- rollout(x0, U) for every item up front (control.py:164)
        prof.begin(3, -1, st);
        rc = Passes<R>::forward(D, kModeRollout, x0, X, U, nullptr, nullptr, nullptr, 1, nullptr, nullptr, S.J_star, S, nullptr,
                                nullptr, D.B, st);
        if (rc) return bail(rc);
        prof.end(st);
        hipLaunchKernelGGL(k_copy_f64, dim3((D.B + 255) / 256), dim3(256), 0, st, D.B, S.J_star, S.J_last);
        if (n_lqr_iter > 0) {
            // tile records: the producer of the loop writes only structurally non-zero entries, so the zeros are put in
            // place once; the big sweep's scratch: its padding is zero and never written
            if (!fused && hipMemsetAsync(tiles, 0, W.K - W.tiles, st) != hipSuccess) return bail(fail(DPILQR_EHIP, "hipMemsetAsync failed"));
            // the static part is written once into all Wn slots here (with items 0..Wn-1 as stand-ins; their (X, U)
            // dependent entries are overwritten by each iteration's producer launch) and skipped afterwards
            if (static_part) {
                if constexpr (sizeof(R) == 8) {
                    if ((rc = launch_make_tiles(D, X, U, tiles, nullptr, nullptr, Wn, true, false, st))) return bail(rc);
                }
            }
        }
    }

    // Iteration loop with continuous admission.  At most Wn sub-problems are in flight; one global iteration = one
    // backward pass + one line search for every active item.  Items that finish are retired by the line-search kernel
    // (it pushes only the survivors onto the next list) and k_admit refills their places from the not-yet-started items,
    // so every launch stays at Wn items although the items need very different numbers of iterations.  The active set
    // lives on the device; the host launches Wn-wide grids (surplus workgroups exit at once).
    size_t n_iterations = 0;
    auto enqueue_iteration = [&](int it, int upper) -> int32_t {
        int32_t* cur = lists + (size_t)(it & 1) * Wn;
        int32_t* cur_n = counts + (it % kCountRing);
        int32_t* nxt_n = counts + ((it + 1) % kCountRing);
        int32_t* mail = solver ? solver->dev + kMailWords * (it % kMailRing) : nullptr;
        hipLaunchKernelGGL(k_admit, dim3(1), dim3(256), 0, st, cur, cur_n, nxt_n, admitted_dev, D.B, Wn, mail,
                           (solver && solver->progress) ? 1 : 0, t_admit, S.fault);
        if (solver) HIP_TRY(hipEventRecord(solver->ev[it % kMailRing], st));
        S.next_items = lists + (size_t)((it + 1) & 1) * Wn;
        S.next_count = nxt_n;
        int32_t r = DPILQR_OK;
        if (fused) {
            if constexpr (sizeof(R) == 8) {
                prof.begin(1, it, st);
                if ((r = launch_riccati_fused(D, X, U, S.mu, K, d, singular, cur, cur_n, upper, S.gains_by_item, st))) return r;
                prof.end(st, g_sweep_waves);
            }
        } else if (!sh.big) {
            if constexpr (sizeof(R) == 8) {
                prof.begin(0, it, st);
                if ((r = launch_make_tiles(D, X, U, tiles, cur, cur_n, upper, true, static_part, st))) return r;
                prof.end(st);
                prof.begin(1, it, st);
                if ((r = launch_riccati(D.B, D.T, n, m, tiles, S.mu, K, d, singular, cur, cur_n, upper, S.gains_by_item, D.n_s,
                                        D.n_c, st)))
                    return r;
                prof.end(st, g_sweep_waves);
            }
        } else {
            prof.begin(1, it, st);
            if ((r = Passes<R>::sweep_big(D, X, U, S.mu, K, d, singular, cur, cur_n, upper, S.gains_by_item, tiles, st))) return r;
            prof.end(st, 0);
        }
        prof.begin(2, it, st);
        if ((r = Passes<R>::forward(D, kModeLineSearch, nullptr, X, U, K, d, alphas, DPILQR_N_ALPHA, Xc, Uc, nullptr, S, cur, cur_n,
                                    upper, st)))
            return r;
        prof.end(st);
        return DPILQR_OK;
    };

    if (n_lqr_iter > 0 && solver) {
        // the host reads {active, admitted} kHostLag iterations late -- that many iterations of launches are always
        // queued while it waits -- to learn when everything has been started and nothing is left active
        int upper = Wn;
        int32_t reported = 0;
        for (int it = 0; it < kMaxGlobalIter; ++it) {
            n_iterations = (size_t)it + 1;
            if ((rc = enqueue_iteration(it, upper))) return bail(rc);
            bool done = false;
            if (it >= kHostLag) {  // {active, admitted} of iteration it - kHostLag
                const int slot = (it - kHostLag) % kMailRing;
                if (hipEventSynchronize(solver->ev[slot]) != hipSuccess) return bail(fail(DPILQR_EHIP, "hipEventSynchronize failed"));
                const int32_t act = solver->host[kMailWords * slot], adm = solver->host[kMailWords * slot + 1];
                const int32_t prefix = solver->host[kMailWords * slot + 2];
                solver->hist.push_back(act);
                // items below `prefix` had finished when that iteration began, and the event says its launches have been
                // reached: their results are final in memory -- the caller may start moving them (on another stream)
                if (solver->progress && prefix > reported) { reported = prefix; solver->progress(solver->progress_user, prefix, D.B); }
                // everything started and the list already empty back then: the iterations since were no-ops
                done = (adm >= D.B && act == 0);
                // once everything is admitted the list can only shrink: tighten the grid
                upper = (adm >= D.B) ? std::min(Wn, std::max(act, 1)) : Wn;
            }
            if (done) break;
            if (it + 1 == kMaxGlobalIter) return bail(fail(DPILQR_EUNSUPPORTED, "solve_batch: more than %d global iterations", kMaxGlobalIter));
        }
    } else if (n_lqr_iter > 0) {
        for (int it = 0; it < n_global_iter; ++it)
            if ((rc = enqueue_iteration(it, Wn))) return rc;
        if (n_global_iter > 0)
            hipLaunchKernelGGL(k_normalise_lists, dim3(1), dim3(256), 0, st, lists, counts, Wn, n_global_iter);
    }
    hipLaunchKernelGGL(k_copy_f64, dim3((D.B + 255) / 256), dim3(256), 0, st, D.B, S.J_last, J);
    if (n_lqr_iter == 0) hipLaunchKernelGGL(k_finish_status, dim3((D.B + 255) / 256), dim3(256), 0, st, D.B, status);
    HIP_TRY(hipGetLastError());
    if (!solver) return DPILQR_OK;       // enqueue-only: everything is on the stream, nothing was waited for
    if (hipStreamSynchronize(st) != hipSuccess) return bail(fail(DPILQR_EHIP, "hipStreamSynchronize failed"));
    if (n_lqr_iter > 0) {
        // the last iterations' exact lengths were posted but not yet consumed
        for (size_t j = solver->hist.size(); j < n_iterations; ++j) solver->hist.push_back(solver->host[kMailWords * (j % kMailRing)]);
    }
    prof.collect(solver->hist, D.B);
    if (solver->progress) solver->progress(solver->progress_user, D.B, D.B);   // the stream has been waited for: everything is final
    if (n_lqr_iter > 0) {
        // items the device gave up (DPILQR_STATUS_FAULT): the count only grows, and the iterations since the list emptied were
        // no-ops, so the freshest mailbox holds the final figure
        int32_t faulted = 0;
        for (int j = 0; j < kMailRing; ++j) faulted = std::max(faulted, solver->host[kMailWords * j + 3]);
        if (faulted > 0)
            return fail(DPILQR_EHIP, "solve_batch: the device gave up %d of %d items (status DPILQR_STATUS_FAULT: a hand-over inside a "
                                     "team of workgroups expired); every other item's result is valid", faulted, D.B);
    }
    return DPILQR_OK;
}

}  // namespace

extern "C" {

int32_t dpilqr_abi_version(void) { return DPILQR_ABI_VERSION; }
const char* dpilqr_last_error(void) { return last_error(); }

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
    return launch_model_op(op, n, ns, model, x, u, dt, o1, o2, as_stream(stream));
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
    if ((int64_t)desc->B * n_pts == 0) return DPILQR_OK;
    return launch_cost_eval(*desc, n_pts, x, u, terminal, cost, as_stream(stream));
}

int32_t dpilqr_tile_layout(int32_t n_x, int32_t n_u, int64_t offsets[7], int64_t row_strides[7], int64_t* stride) {
    if (n_x < 1 || n_u < 1 || !offsets || !row_strides || !stride) return fail(DPILQR_EINVAL, "tile_layout: bad argument");
    return tile_layout_host(n_x, n_u, offsets, row_strides, stride);
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

int64_t dpilqr_backward_pass_workspace_bytes(const dpilqr_batch_desc* desc, int32_t elem_bytes) {
    if (!desc || desc->B < 0 || desc->k < 1 || desc->T < 1 || (elem_bytes != 4 && elem_bytes != 8))
        return fail(DPILQR_EINVAL, "backward_pass_workspace_bytes: bad argument");
    const int n = desc->k * desc->n_s, m = desc->k * desc->n_c;
    if (elem_bytes == 4 || uses_big_path(n)) return (int64_t)elem_bytes * desc->B * riccati_big_scratch_elems(n, m);
    return (int64_t)sizeof(double) * desc->B * (desc->T + 1) * TileLayout(n, m).stride;
}

int32_t dpilqr_backward_pass(const dpilqr_batch_desc* desc, const double* X, const double* U, const double* mu,
                             double* K, double* d, double* tiles_workspace, void* stream) {
    int32_t rc = check_desc(desc);
    if (rc) return rc;
    if (desc->B == 0) return DPILQR_OK;
    if (!X || !U || !mu || !K || !d || !tiles_workspace) return fail(DPILQR_EINVAL, "backward_pass: NULL pointer");
    const int n = desc->k * desc->n_s, m = desc->k * desc->n_c;
    if (uses_big_path(n)) {   // large clusters: the fused sweep (no tile records); the workspace is its scratch
        HIP_TRY(hipMemsetAsync(tiles_workspace, 0, sizeof(double) * desc->B * (size_t)riccati_big_scratch_elems(n, m), as_stream(stream)));
        return launch_riccati_big_f64(*desc, X, U, mu, K, d, nullptr, nullptr, nullptr, desc->B, 0, tiles_workspace, as_stream(stream));
    }
    rc = launch_make_tiles(*desc, X, U, tiles_workspace, nullptr, nullptr, desc->B, false, false, as_stream(stream));
    if (rc) return rc;
    return launch_riccati(desc->B, desc->T, n, m, tiles_workspace, mu, K, d, nullptr, nullptr, nullptr, desc->B, 0, desc->n_s,
                          desc->n_c, as_stream(stream));
}

int32_t dpilqr_backward_pass_fused(const dpilqr_batch_desc* desc, const double* X, const double* U, const double* mu, double* K,
                                   double* d, int32_t* singular, void* stream) {
    int32_t rc = check_desc(desc);
    if (rc) return rc;
    if (desc->B == 0) return DPILQR_OK;
    if (!X || !U || !mu || !K || !d) return fail(DPILQR_EINVAL, "backward_pass_fused: NULL pointer");
    if (!fused_sweep_applies(*desc))
        return fail(DPILQR_EUNSUPPORTED, "backward_pass_fused: needs 6..15 four-state, 1..10 six-state or 1..6 CarDynamics3D agents, or at most five "
                                         "DoubleIntDynamics4D / UnicycleDynamics4D agents with n_dims = 2 (uniform_model hints)");
    rc = launch_riccati_fused(*desc, X, U, mu, K, d, singular, nullptr, nullptr, desc->B, 0, as_stream(stream));
    return rc == DPILQR_EUNSUPPORTED ? fail(rc, "backward_pass_fused: no instantiation for n_x=%d", desc->k * desc->n_s) : rc;
}

int32_t dpilqr_backward_pass_f32(const dpilqr_batch_desc* desc, const float* X, const float* U, const double* mu, float* K,
                                 float* d, void* workspace, void* stream) {
    int32_t rc = check_desc(desc);
    if (rc) return rc;
    if (desc->B == 0) return DPILQR_OK;
    if (!X || !U || !mu || !K || !d || !workspace) return fail(DPILQR_EINVAL, "backward_pass_f32: NULL pointer");
    const int n = desc->k * desc->n_s, m = desc->k * desc->n_c;
    HIP_TRY(hipMemsetAsync(workspace, 0, sizeof(float) * desc->B * (size_t)riccati_big_scratch_elems(n, m), as_stream(stream)));
    return launch_riccati_big_f32(*desc, X, U, mu, K, d, nullptr, nullptr, nullptr, desc->B, 0, workspace, as_stream(stream));
}

int32_t dpilqr_rollout_f32(const dpilqr_batch_desc* desc, const float* x0, const float* U, float* X, double* J, void* stream) {
    int32_t rc = check_desc(desc);
    if (rc) return rc;
    if (desc->B == 0) return DPILQR_OK;
    if (!x0 || !U || !X || !J) return fail(DPILQR_EINVAL, "rollout_f32: NULL pointer");
    SolveState S{};
    return launch_forward_big_f32(*desc, kModeRollout, x0, X, const_cast<float*>(U), nullptr, nullptr, nullptr, 1, nullptr, nullptr,
                                  J, S, nullptr, nullptr, desc->B, as_stream(stream));
}

int32_t dpilqr_forward_pass_f32(const dpilqr_batch_desc* desc, const float* X, const float* U, const float* K, const float* d,
                                const double* alphas, int32_t n_alpha, float* Xn, float* Un, double* Jn, void* stream) {
    int32_t rc = check_desc(desc);
    if (rc) return rc;
    if (desc->B == 0) return DPILQR_OK;
    if (!X || !U || !K || !d || !alphas || !Xn || !Un || !Jn) return fail(DPILQR_EINVAL, "forward_pass_f32: NULL pointer");
    if (n_alpha < 1) return fail(DPILQR_EINVAL, "forward_pass_f32: n_alpha=%d", n_alpha);
    SolveState S{};
    return launch_forward_big_f32(*desc, kModeCandidates, nullptr, const_cast<float*>(X), const_cast<float*>(U), K, d, alphas,
                                  n_alpha, Xn, Un, Jn, S, nullptr, nullptr, desc->B, as_stream(stream));
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
    return (int64_t)SolveWorkspace(*desc, window_of(*desc, window), gains_in_workspace != 0, shape_of<double>(*desc)).total;
}
int64_t dpilqr_solve_workspace_bytes_f32(const dpilqr_batch_desc* desc, int32_t window, int32_t gains_in_workspace) {
    if (!desc || desc->B < 0 || desc->k < 1 || desc->T < 1) return fail(DPILQR_EINVAL, "solve_workspace_bytes_f32: bad desc");
    return (int64_t)SolveWorkspace(*desc, window_of(*desc, window), gains_in_workspace != 0, shape_of<float>(*desc)).total;
}

int32_t dpilqr_solver_create(dpilqr_solver** out) {
    if (!out) return fail(DPILQR_EINVAL, "solver_create: NULL pointer");
    dpilqr_solver* s = new (std::nothrow) dpilqr_solver();
    if (!s) return fail(DPILQR_EHIP, "solver_create: out of host memory");
    const int32_t rc = s->init();
    if (rc) { delete s; return rc; }
    *out = s;
    return DPILQR_OK;
}
int32_t dpilqr_solver_destroy(dpilqr_solver* solver) {
    delete solver;
    return DPILQR_OK;
}
int32_t dpilqr_solver_set_progress(dpilqr_solver* solver, dpilqr_progress_fn fn, void* user) {
    if (!solver) {
        const int32_t rc = default_solver(&solver);
        if (rc) return rc;
    }
    solver->progress = fn;
    solver->progress_user = user;
    return DPILQR_OK;
}

int32_t dpilqr_solve_batch(dpilqr_solver* solver, const dpilqr_batch_desc* desc, const double* x0, double* U,
                           int32_t n_lqr_iter, double tol, double t_kill, int32_t window, void* workspace, int64_t workspace_bytes,
                           double* X, double* J, int32_t* status, int32_t* n_bwd, int32_t* n_fwd, double* trace,
                           double* K_out, double* d_out, void* stream) {
    if (!solver) {
        const int32_t rc = default_solver(&solver);
        if (rc) return rc;
    }
    return solve_impl<double>(solver, desc, x0, U, n_lqr_iter, tol, t_kill, window, workspace, workspace_bytes, X, J, status, n_bwd,
                              n_fwd, trace, K_out, d_out, 0, 0, as_stream(stream));
}
int32_t dpilqr_solve_batch_f32(dpilqr_solver* solver, const dpilqr_batch_desc* desc, const float* x0, float* U,
                               int32_t n_lqr_iter, double tol, double t_kill, int32_t window, void* workspace, int64_t workspace_bytes,
                               float* X, double* J, int32_t* status, int32_t* n_bwd, int32_t* n_fwd, double* trace,
                               float* K_out, float* d_out, void* stream) {
    if (!solver) {
        const int32_t rc = default_solver(&solver);
        if (rc) return rc;
    }
    return solve_impl<float>(solver, desc, x0, U, n_lqr_iter, tol, t_kill, window, workspace, workspace_bytes, X, J, status, n_bwd,
                             n_fwd, trace, K_out, d_out, 0, 0, as_stream(stream));
}
int32_t dpilqr_solve_enqueue(const dpilqr_batch_desc* desc, const double* x0, double* U, int32_t n_lqr_iter, double tol,
                             double t_kill, int32_t window, void* workspace, int64_t workspace_bytes, double* X, double* J, int32_t* status,
                             int32_t* n_bwd, int32_t* n_fwd, double* trace, double* K_out, double* d_out,
                             int32_t n_global_iter, int32_t resume, void* stream) {
    return solve_impl<double>(nullptr, desc, x0, U, n_lqr_iter, tol, t_kill, window, workspace, workspace_bytes, X, J, status, n_bwd,
                              n_fwd, trace, K_out, d_out, n_global_iter, resume, as_stream(stream));
}
int64_t dpilqr_solve_iterations_bound(const dpilqr_batch_desc* desc, int32_t window, int32_t n_lqr_iter) {
    if (!desc || desc->B < 0 || n_lqr_iter < 0) return fail(DPILQR_EINVAL, "solve_iterations_bound: bad argument");
    const int64_t W = window_of(*desc, window);
    // every global iteration with a non-empty list completes one iLQR iteration of >= 1 item, and an item that is in
    // flight stays in flight until it finishes: ceil(B / W) generations of at most n_lqr_iter iterations each
    return ((int64_t)desc->B + W - 1) / W * n_lqr_iter + 1;
}

int32_t dpilqr_debug_stamps(void* buf) {
    int32_t rc = set_stamp_buffer_riccati(buf);
    return rc ? rc : set_stamp_buffer_forward(buf);
}

int32_t dpilqr_profile_enable(dpilqr_solver* solver, int32_t enable) {
    Profiler& g_prof = solver ? solver->prof : g_default_solver.prof;
    const int32_t prev = g_prof.on ? 1 : 0;
    g_prof.on = (enable & 1) != 0;
    g_prof.mask = (enable >> 1) & 0xF ? (enable >> 1) & 0xF : 0xF;
    return prev;
}

int32_t dpilqr_profile_read(dpilqr_solver* solver, double ms[4], int64_t launches[4], int64_t items[4], int32_t reset) {
    if (!ms || !launches || !items) return fail(DPILQR_EINVAL, "profile_read: NULL pointer");
    Profiler& g_prof = solver ? solver->prof : g_default_solver.prof;
    for (int c = 0; c < 4; ++c) {
        ms[c] = g_prof.ms[c]; launches[c] = g_prof.launches[c]; items[c] = g_prof.items[c];
        if (reset) { g_prof.ms[c] = 0; g_prof.launches[c] = 0; g_prof.items[c] = 0; }
    }
    return DPILQR_OK;
}

int32_t dpilqr_profile_read_sweep(dpilqr_solver* solver, int32_t waves, double* ms, int64_t* launches, int64_t* items, int32_t reset) {
    if (!ms || !launches || !items) return fail(DPILQR_EINVAL, "profile_read_sweep: NULL pointer");
    if (waves != 4 && waves != 8 && waves != 12) return fail(DPILQR_EINVAL, "profile_read_sweep: waves=%d (4, 8 or 12)", waves);
    const int v = waves / 4 - 1;
    Profiler& g_prof = solver ? solver->prof : g_default_solver.prof;
    *ms = g_prof.sweep_ms[v]; *launches = g_prof.sweep_launches[v]; *items = g_prof.sweep_items[v];
    if (reset) { g_prof.sweep_ms[v] = 0; g_prof.sweep_launches[v] = 0; g_prof.sweep_items[v] = 0; }
    return DPILQR_OK;
}

int32_t dpilqr_pairwise_graph(int32_t S, int32_t N, int32_t k, int32_t n_s, const double* X, const double* radius,
                              int32_t* adj, void* stream) {
    if (S < 0 || N < 1 || k < 1 || n_s < 2 || !X || !radius || !adj) return fail(DPILQR_EINVAL, "pairwise_graph: bad argument");
    if (S == 0) return DPILQR_OK;
    return launch_pairwise_graph(S, N, k, n_s, X, radius, adj, as_stream(stream));
}

}  // extern "C"
