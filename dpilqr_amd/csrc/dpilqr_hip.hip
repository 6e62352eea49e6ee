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
    int32_t rc = set_stamp_buffer_riccati(buf);
    return rc ? rc : set_stamp_buffer_forward(buf);
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
    return launch_pairwise_graph(S, N, k, n_s, X, radius, adj, as_stream(stream));
}

}  // extern "C"
