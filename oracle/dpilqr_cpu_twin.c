/*
 * dpilqr_cpu_twin.c -- TEST INFRASTRUCTURE ONLY: the C ABI of include/dpilqr_hip.h over the CPU oracle.
 *
 * SURVEY.md 8(b) asks for "an identical-ABI CPU build (same symbols, host pointers) [that] serves as baseline and
 * debugger".  This is it: libdpilqr_cpu_twin.so exports EVERY symbol the header declares, takes HOST pointers where
 * the HIP library takes device pointers, ignores `stream`, and computes with oracle/ilqr_oracle.c.  The entry points
 * of the hot path (model FFI, cost, tiles, rollout, backward / forward pass, the whole solve, the interaction graph) are
 * implemented; the ones that only make sense on the device or are measurement hooks (fp32 arm, fused / enqueue
 * variants, dispatch front end, scenario generation, profiler) return DPILQR_EUNSUPPORTED.
 *
 * It is NOT a fallback of the product and cannot become one by accident: dpilqr_device_info always answers DPILQR_ENOGPU,
 * so dpilqr_amd._lib.require_gpu() refuses it, and nothing under dpilqr_amd/ ever names this file.  Only
 * tests/test_cpu_twin.py loads it (to check the ABI's data layouts -- batch descriptor strides, tile records, item lists,
 * statuses, traces -- against the golden vectors without a GPU, and as a debugger's second opinion).
 */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/dpilqr_hip.h"
#include "ilqr_oracle.h"

static __thread char g_err[256] = "";
static int32_t fail(int32_t code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define UNSUPPORTED(name) return fail(DPILQR_EUNSUPPORTED, name ": not part of the CPU twin")

static int even(int x) { return (x + 1) & ~1; }
typedef struct { int oA, oB, ldAB, oLxx, oLuu, oLux, ldUG, oLx, oLu, stride; } layout_t;
static layout_t layout_of(int n, int m)
{   /* csrc/tiles.hpp TileLayout */
    layout_t L;
    L.ldAB = n + m; L.oA = 0; L.oB = n; L.oLxx = even(n * L.ldAB); L.ldUG = m + n; L.oLuu = even(L.oLxx + n * n);
    L.oLux = L.oLuu + m; L.oLx = even(L.oLuu + m * L.ldUG); L.oLu = L.oLx + n; L.stride = even(L.oLu + m);
    return L;
}

static oracle_problem item_problem(const dpilqr_batch_desc *D, int b)
{
    oracle_problem p;
    p.k = D->k; p.n_s = D->n_s; p.n_c = D->n_c; p.T = D->T; p.dt = D->dt;
    p.model = (const int *)(D->model + (int64_t)b * D->model_bstride);
    p.n_dims = (const int *)(D->n_dims + (int64_t)b * D->n_dims_bstride);
    p.xf = D->xf + (int64_t)b * D->xf_bstride; p.Q = D->Q + (int64_t)b * D->Q_bstride;
    p.R = D->R + (int64_t)b * D->R_bstride; p.Qf = D->Qf + (int64_t)b * D->Qf_bstride;
    p.radius = D->radius[(int64_t)b * D->radius_bstride]; p.w_ref = D->w_ref; p.w_prox = D->w_prox;
    return p;
}
static int32_t check_desc(const dpilqr_batch_desc *d)
{
    if (!d) return fail(DPILQR_EINVAL, "desc is NULL");
    if (d->B < 0 || d->k < 1 || d->T < 1) return fail(DPILQR_EINVAL, "bad sizes");
    return DPILQR_OK;
}

int32_t dpilqr_abi_version(void) { return DPILQR_ABI_VERSION; }
const char *dpilqr_last_error(void) { return g_err; }
int32_t dpilqr_device_info(int32_t dev, int32_t *n_cu, int32_t *lds, char *arch, int32_t arch_len)
{
    (void)dev; (void)n_cu; (void)lds;
    if (arch && arch_len > 0) { strncpy(arch, "cpu-twin", arch_len - 1); arch[arch_len - 1] = 0; }
    return fail(DPILQR_ENOGPU, "this is the CPU twin of libdpilqr_hip.so (test infrastructure): there is no device");
}
int32_t dpilqr_model_dims(int32_t model, int32_t *n_s, int32_t *n_c)
{
    int a, b;
    if (!n_s || !n_c || oracle_model_dims(model, &a, &b)) return fail(DPILQR_EINVAL, "unknown model %d", model);
    *n_s = a; *n_c = b;
    return DPILQR_OK;
}
static int family_nc(int ns) { return ns == 3 ? 2 : ns == 4 ? 2 : ns == 6 ? 3 : ns == 12 ? 4 : -1; }
int32_t dpilqr_model_f(int32_t n, int32_t ns, const int32_t *model, const double *x, const double *u, double *o, void *s)
{
    (void)s; const int nc = family_nc(ns);
    if (n < 0 || nc < 0 || !model || !x || !u || !o) return fail(DPILQR_EINVAL, "model op: bad argument");
    for (int i = 0; i < n; ++i) oracle_model_f(model[i], x + (size_t)i * ns, u + (size_t)i * nc, o + (size_t)i * ns);
    return DPILQR_OK;
}
int32_t dpilqr_model_integrate(int32_t n, int32_t ns, const int32_t *model, const double *x, const double *u, double dt, double *o, void *s)
{
    (void)s; const int nc = family_nc(ns);
    if (n < 0 || nc < 0 || !model || !x || !u || !o) return fail(DPILQR_EINVAL, "model op: bad argument");
    for (int i = 0; i < n; ++i) oracle_model_integrate(model[i], x + (size_t)i * ns, u + (size_t)i * nc, dt, o + (size_t)i * ns);
    return DPILQR_OK;
}
int32_t dpilqr_model_linearize(int32_t n, int32_t ns, const int32_t *model, const double *x, const double *u, double dt, double *A, double *B, void *s)
{
    (void)s; const int nc = family_nc(ns);
    if (n < 0 || nc < 0 || !model || !x || !u || !A || !B) return fail(DPILQR_EINVAL, "model op: bad argument");
    for (int i = 0; i < n; ++i)
        oracle_model_linearize(model[i], x + (size_t)i * ns, u + (size_t)i * nc, dt, A + (size_t)i * ns * ns, B + (size_t)i * ns * nc);
    return DPILQR_OK;
}
int32_t dpilqr_cost_eval(const dpilqr_batch_desc *D, int32_t n_pts, const double *x, const double *u, int32_t terminal, double *cost, void *s)
{
    (void)s;
    if (check_desc(D) || n_pts < 0 || !x || !u || !cost) return fail(DPILQR_EINVAL, "cost_eval: bad argument");
    const int n = D->k * D->n_s, m = D->k * D->n_c;
    for (int b = 0; b < D->B; ++b) {
        const oracle_problem p = item_problem(D, b);
        for (int i = 0; i < n_pts; ++i) {
            const size_t e = (size_t)b * n_pts + i;
            cost[e] = oracle_cost(&p, x + e * n, u + e * m, terminal);
        }
    }
    return DPILQR_OK;
}
int32_t dpilqr_tile_layout(int32_t n, int32_t m, int64_t off[7], int64_t ld[7], int64_t *stride)
{
    if (n < 1 || m < 1 || !off || !ld || !stride) return fail(DPILQR_EINVAL, "tile_layout: bad argument");
    const layout_t L = layout_of(n, m);
    off[0] = L.oA; off[1] = L.oB; off[2] = L.oLxx; off[3] = L.oLux; off[4] = L.oLuu; off[5] = L.oLx; off[6] = L.oLu;
    ld[0] = L.ldAB; ld[1] = L.ldAB; ld[2] = n; ld[3] = L.ldUG; ld[4] = L.ldUG; ld[5] = 1; ld[6] = 1;
    *stride = L.stride;
    return DPILQR_OK;
}
int64_t dpilqr_tiles_bytes(int32_t B, int32_t T, int32_t n, int32_t m)
{
    if (B < 0 || T < 1 || n < 1 || m < 1) return fail(DPILQR_EINVAL, "tiles_bytes: bad argument");
    return (int64_t)sizeof(double) * B * (T + 1) * layout_of(n, m).stride;
}
int32_t dpilqr_make_tiles(const dpilqr_batch_desc *D, const double *X, const double *U, double *tiles, const int32_t *items,
                          const int32_t *n_items, void *s)
{
    (void)s;
    if (check_desc(D) || !X || !U || !tiles) return fail(DPILQR_EINVAL, "make_tiles: bad argument");
    const int n = D->k * D->n_s, m = D->k * D->n_c, T = D->T;
    const layout_t L = layout_of(n, m);
    const int cnt = n_items ? *n_items : D->B;
    double *A = malloc(sizeof(double) * (2 * n * n + n * m + m * m + m * n + n + m)), *Bm = A + n * n, *Lxx = Bm + n * m,
           *Luu = Lxx + n * n, *Lux = Luu + m * m, *Lx = Lux + m * n, *Lu = Lx + n;
    double *uz = calloc(m, sizeof(double));
    for (int slot = 0; slot < cnt; ++slot) {
        const int b = items ? items[slot] : slot;
        const oracle_problem p = item_problem(D, b);
        for (int t = 0; t <= T; ++t) {
            double *rec = tiles + ((size_t)slot * (T + 1) + t) * L.stride;
            const double *x = X + ((size_t)b * (T + 1) + t) * n, *u = t < T ? U + ((size_t)b * T + t) * m : uz;
            memset(rec, 0, sizeof(double) * L.stride);
            oracle_quadraticize(&p, x, u, t == T, Lx, Lu, Lxx, Luu, Lux);
            if (t < T) {
                oracle_linearize(&p, x, u, A, Bm);
                for (int i = 0; i < n; ++i) {
                    memcpy(rec + L.oA + i * L.ldAB, A + i * n, sizeof(double) * n);
                    memcpy(rec + L.oB + i * L.ldAB, Bm + i * m, sizeof(double) * m);
                }
                for (int a = 0; a < m; ++a) {
                    memcpy(rec + L.oLuu + a * L.ldUG, Luu + a * m, sizeof(double) * m);
                    memcpy(rec + L.oLux + a * L.ldUG, Lux + a * n, sizeof(double) * n);
                }
                memcpy(rec + L.oLu, Lu, sizeof(double) * m);
            }
            memcpy(rec + L.oLxx, Lxx, sizeof(double) * n * n);
            memcpy(rec + L.oLx, Lx, sizeof(double) * n);
        }
    }
    free(A); free(uz);
    return DPILQR_OK;
}
int32_t dpilqr_rollout(const dpilqr_batch_desc *D, const double *x0, const double *U, double *X, double *J, void *s)
{
    (void)s;
    if (check_desc(D) || !x0 || !U || !X || !J) return fail(DPILQR_EINVAL, "rollout: bad argument");
    const int n = D->k * D->n_s, m = D->k * D->n_c, T = D->T;
#pragma omp parallel for
    for (int b = 0; b < D->B; ++b) {
        const oracle_problem p = item_problem(D, b);
        J[b] = oracle_rollout(&p, x0 + (size_t)b * n, U + (size_t)b * T * m, X + (size_t)b * (T + 1) * n);
    }
    return DPILQR_OK;
}
int32_t dpilqr_backward_pass_tiles_blocks(int32_t B, int32_t T, int32_t n, int32_t m, int32_t bns, int32_t bnc, const double *tiles,
                                          const double *mu, double *K, double *d, int32_t *singular, const int32_t *items,
                                          const int32_t *n_items, void *s)
{
    (void)s; (void)bns; (void)bnc;   /* the block promise only lets the device skip zero products */
    if (B < 0 || T < 1 || n < 1 || m < 1 || !tiles || !mu || !K || !d) return fail(DPILQR_EINVAL, "backward_pass_tiles: bad argument");
    const layout_t L = layout_of(n, m);
    const int cnt = n_items ? *n_items : B;
#pragma omp parallel for
    for (int slot = 0; slot < cnt; ++slot) {
        const int b = items ? items[slot] : slot;
        double *A = malloc(sizeof(double) * ((size_t)T * (n * n + n * m) + (size_t)(T + 1) * (n + m + n * n + m * m + m * n)));
        double *Bm = A + (size_t)T * n * n, *Lx = Bm + (size_t)T * n * m, *Lu = Lx + (size_t)(T + 1) * n, *Lxx = Lu + (size_t)(T + 1) * m,
               *Luu = Lxx + (size_t)(T + 1) * n * n, *Lux = Luu + (size_t)(T + 1) * m * m;
        for (int t = 0; t <= T; ++t) {
            const double *rec = tiles + ((size_t)slot * (T + 1) + t) * L.stride;
            if (t < T)
                for (int i = 0; i < n; ++i) {
                    memcpy(A + ((size_t)t * n + i) * n, rec + L.oA + i * L.ldAB, sizeof(double) * n);
                    memcpy(Bm + ((size_t)t * n + i) * m, rec + L.oB + i * L.ldAB, sizeof(double) * m);
                }
            for (int a = 0; a < m; ++a) {
                memcpy(Luu + ((size_t)t * m + a) * m, rec + L.oLuu + a * L.ldUG, sizeof(double) * m);
                memcpy(Lux + ((size_t)t * m + a) * n, rec + L.oLux + a * L.ldUG, sizeof(double) * n);
            }
            memcpy(Lxx + (size_t)t * n * n, rec + L.oLxx, sizeof(double) * n * n);
            memcpy(Lx + (size_t)t * n, rec + L.oLx, sizeof(double) * n);
            memcpy(Lu + (size_t)t * m, rec + L.oLu, sizeof(double) * m);
        }
        const int rc = oracle_backward_pass_tiles(n, m, T, A, Bm, Lx, Lu, Lxx, Luu, Lux, mu[b], K + (size_t)slot * T * m * n,
                                                  d + (size_t)slot * T * m);
        if (rc && singular) singular[b] = 1;
        free(A);
    }
    return DPILQR_OK;
}
int32_t dpilqr_backward_pass_tiles(int32_t B, int32_t T, int32_t n, int32_t m, const double *tiles, const double *mu, double *K, double *d,
                                   int32_t *singular, const int32_t *items, const int32_t *n_items, void *s)
{
    return dpilqr_backward_pass_tiles_blocks(B, T, n, m, 0, 0, tiles, mu, K, d, singular, items, n_items, s);
}
int64_t dpilqr_backward_pass_workspace_bytes(const dpilqr_batch_desc *D, int32_t eb) { (void)D; (void)eb; return 8; }
int32_t dpilqr_backward_pass(const dpilqr_batch_desc *D, const double *X, const double *U, const double *mu, double *K, double *d,
                             double *ws, void *s)
{
    (void)ws; (void)s;
    if (check_desc(D) || !X || !U || !mu || !K || !d) return fail(DPILQR_EINVAL, "backward_pass: bad argument");
    const int n = D->k * D->n_s, m = D->k * D->n_c, T = D->T;
#pragma omp parallel for
    for (int b = 0; b < D->B; ++b) {
        const oracle_problem p = item_problem(D, b);
        oracle_backward_pass(&p, X + (size_t)b * (T + 1) * n, U + (size_t)b * T * m, mu[b], K + (size_t)b * T * m * n, d + (size_t)b * T * m);
    }
    return DPILQR_OK;
}
int32_t dpilqr_forward_pass(const dpilqr_batch_desc *D, const double *X, const double *U, const double *K, const double *d,
                            const double *alphas, int32_t na, double *Xn, double *Un, double *Jn, void *s)
{
    (void)s;
    if (check_desc(D) || !X || !U || !K || !d || !alphas || !Xn || !Un || !Jn || na < 1) return fail(DPILQR_EINVAL, "forward_pass: bad argument");
    const int n = D->k * D->n_s, m = D->k * D->n_c, T = D->T;
#pragma omp parallel for
    for (int b = 0; b < D->B; ++b) {
        const oracle_problem p = item_problem(D, b);
        for (int g = 0; g < na; ++g)
            Jn[(size_t)b * na + g] = oracle_forward_pass(&p, X + (size_t)b * (T + 1) * n, U + (size_t)b * T * m, K + (size_t)b * T * m * n,
                                                         d + (size_t)b * T * m, alphas[g], Xn + ((size_t)b * na + g) * (T + 1) * n,
                                                         Un + ((size_t)b * na + g) * T * m);
    }
    return DPILQR_OK;
}
int32_t dpilqr_alphas(double *a)
{
    if (!a) return fail(DPILQR_EINVAL, "alphas: NULL pointer");
    oracle_alphas(a);
    return DPILQR_OK;
}
int64_t dpilqr_solve_workspace_bytes(const dpilqr_batch_desc *D, int32_t w, int32_t g) { (void)D; (void)w; (void)g; return 8; }
struct dpilqr_solver { dpilqr_progress_fn progress; void *progress_user; };
int32_t dpilqr_solver_create(dpilqr_solver **out)
{
    if (!out) return fail(DPILQR_EINVAL, "solver_create: NULL pointer");
    *out = calloc(1, sizeof(dpilqr_solver));
    return DPILQR_OK;
}
int32_t dpilqr_solver_destroy(dpilqr_solver *sv) { free(sv); return DPILQR_OK; }
static _Thread_local dpilqr_solver default_solver;       /* solver = NULL: the calling thread's default solver (header) */
int32_t dpilqr_solver_set_progress(dpilqr_solver *sv, dpilqr_progress_fn fn, void *user)
{
    if (!sv) sv = &default_solver;
    sv->progress = fn; sv->progress_user = user;
    return DPILQR_OK;
}
int32_t dpilqr_solve_batch(dpilqr_solver *sv, const dpilqr_batch_desc *D, const double *x0, double *U, int32_t n_lqr_iter, double tol,
                           double t_kill, int32_t window, void *ws, int64_t ws_bytes, double *X, double *J, int32_t *status, int32_t *n_bwd,
                           int32_t *n_fwd, double *trace, double *K_out, double *d_out, void *s)
{
    (void)window; (void)ws; (void)ws_bytes; (void)s;
    if (t_kill > 0.0) return fail(DPILQR_EUNSUPPORTED, "solve_batch: the CPU twin has no clock-driven bail-out (t_kill); "
                                                       "oracle_solve with n_lqr_iter = the killed item's n_bwd is the same solve");
    if (!sv) sv = &default_solver;
    if (check_desc(D) || !x0 || !U || !X || !J || !status || !n_bwd || !n_fwd) return fail(DPILQR_EINVAL, "solve_batch: bad argument");
    if (K_out || d_out) return fail(DPILQR_EUNSUPPORTED, "solve_batch: the CPU twin does not return the last gains");
    const int n = D->k * D->n_s, m = D->k * D->n_c, T = D->T;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < D->B; ++b) {
        const oracle_problem p = item_problem(D, b);
        int nb = 0, nf = 0;
        status[b] = oracle_solve(&p, x0 + (size_t)b * n, U + (size_t)b * T * m, n_lqr_iter, tol, X + (size_t)b * (T + 1) * n, J + b,
                                 trace ? trace + (size_t)b * (n_lqr_iter > 0 ? n_lqr_iter : 1) * 5 : NULL, &nb, &nf);
        n_bwd[b] = nb; n_fwd[b] = nf;
    }
    if (sv->progress) sv->progress(sv->progress_user, D->B, D->B);      /* the header's "a last time with n_finished = n_items" */
    return DPILQR_OK;
}
int32_t dpilqr_pairwise_graph(int32_t S, int32_t N, int32_t k, int32_t n_s, const double *X, const double *radius, int32_t *adj, void *s)
{
    (void)s;
    if (S < 0 || N < 1 || k < 1 || n_s < 2 || !X || !radius || !adj) return fail(DPILQR_EINVAL, "pairwise_graph: bad argument");
    const int step = (N / 10 > 1) ? N / 10 : 1;
    for (int sc = 0; sc < S; ++sc) {
        int32_t *A = adj + (size_t)sc * k * k;
        memset(A, 0, sizeof(int32_t) * k * k);
        for (int i = 0; i < k; ++i) {
            A[i * k + i] = 1;
            for (int j = i + 1; j < k; ++j)
                for (int r = 0; r < N; r += step) {
                    const double *row = X + ((size_t)sc * N + r) * k * n_s;
                    const double dx = row[i * n_s] - row[j * n_s], dy = row[i * n_s + 1] - row[j * n_s + 1];
                    if (sqrt(dx * dx + dy * dy) < 2 * radius[sc]) { A[i * k + j] = A[j * k + i] = 1; break; }
                }
        }
    }
    return DPILQR_OK;
}

/* ---- declared by the header, device-only or measurement hooks: present (the symbol set is identical), not implemented */
int32_t dpilqr_backward_pass_fused(const dpilqr_batch_desc *D, const double *X, const double *U, const double *mu, double *K, double *d,
                                   int32_t *sg, void *s) { (void)D; (void)X; (void)U; (void)mu; (void)K; (void)d; (void)sg; (void)s; UNSUPPORTED("backward_pass_fused"); }
int32_t dpilqr_solve_enqueue(const dpilqr_batch_desc *D, const double *x0, double *U, int32_t a, double b, double tk, int32_t c, void *ws, int64_t wb,
                             double *X, double *J, int32_t *st, int32_t *nb, int32_t *nf, double *tr, double *K, double *d, int32_t g,
                             int32_t r, void *s)
{ (void)D; (void)x0; (void)U; (void)a; (void)b; (void)tk; (void)c; (void)ws; (void)wb; (void)X; (void)J; (void)st; (void)nb; (void)nf; (void)tr; (void)K; (void)d; (void)g; (void)r; (void)s; UNSUPPORTED("solve_enqueue"); }
int64_t dpilqr_solve_iterations_bound(const dpilqr_batch_desc *D, int32_t w, int32_t n) { (void)D; (void)w; (void)n; UNSUPPORTED("solve_iterations_bound"); }
int32_t dpilqr_rollout_f32(const dpilqr_batch_desc *D, const float *a, const float *b, float *c, double *J, void *s) { (void)D; (void)a; (void)b; (void)c; (void)J; (void)s; UNSUPPORTED("rollout_f32"); }
int32_t dpilqr_backward_pass_f32(const dpilqr_batch_desc *D, const float *a, const float *b, const double *mu, float *K, float *d, void *w, void *s)
{ (void)D; (void)a; (void)b; (void)mu; (void)K; (void)d; (void)w; (void)s; UNSUPPORTED("backward_pass_f32"); }
int32_t dpilqr_forward_pass_f32(const dpilqr_batch_desc *D, const float *a, const float *b, const float *K, const float *d, const double *al,
                                int32_t na, float *Xn, float *Un, double *Jn, void *s)
{ (void)D; (void)a; (void)b; (void)K; (void)d; (void)al; (void)na; (void)Xn; (void)Un; (void)Jn; (void)s; UNSUPPORTED("forward_pass_f32"); }
int64_t dpilqr_solve_workspace_bytes_f32(const dpilqr_batch_desc *D, int32_t w, int32_t g) { (void)D; (void)w; (void)g; UNSUPPORTED("solve_workspace_bytes_f32"); }
int32_t dpilqr_solve_batch_f32(dpilqr_solver *sv, const dpilqr_batch_desc *D, const float *x0, float *U, int32_t a, double b, double tk, int32_t c, void *ws,
                               int64_t wb, float *X, double *J, int32_t *st, int32_t *nb, int32_t *nf, double *tr, float *K, float *d, void *s)
{ (void)sv; (void)D; (void)x0; (void)U; (void)a; (void)b; (void)tk; (void)c; (void)ws; (void)wb; (void)X; (void)J; (void)st; (void)nb; (void)nf; (void)tr; (void)K; (void)d; (void)s; UNSUPPORTED("solve_batch_f32"); }
int32_t dpilqr_profile_enable(dpilqr_solver *sv, int32_t e) { (void)sv; (void)e; return 0; }
int32_t dpilqr_debug_stamps(void *b) { (void)b; UNSUPPORTED("debug_stamps"); }
int32_t dpilqr_profile_read(dpilqr_solver *sv, double ms[4], int64_t l[4], int64_t it[4], int32_t r) { (void)sv; (void)ms; (void)l; (void)it; (void)r; UNSUPPORTED("profile_read"); }
int32_t dpilqr_profile_read_sweep(dpilqr_solver *sv, int32_t w, double *ms, int64_t *l, int64_t *it, int32_t r) { (void)sv; (void)w; (void)ms; (void)l; (void)it; (void)r; UNSUPPORTED("profile_read_sweep"); }
int32_t dpilqr_dispatch_graph(int32_t S, int32_t N, int32_t k, int32_t ns, const double *X, const double *r, int64_t rs, const int32_t *ig, uint64_t *bits,
                              int32_t *rep, int32_t *size, int32_t *order, int32_t *slot, int32_t *bs, int32_t *bc, void *s)
{ (void)S; (void)N; (void)k; (void)ns; (void)X; (void)r; (void)rs; (void)ig; (void)bits; (void)rep; (void)size; (void)order; (void)slot; (void)bs; (void)bc; (void)s; UNSUPPORTED("dispatch_graph"); }
int32_t dpilqr_dispatch_gather(int32_t k, int32_t ns, int32_t nc, int32_t T, int32_t nr, int32_t kc, const int32_t *o, int32_t f, int32_t c, const uint64_t *b,
                               const double *X, const double *U, const double *xf, int64_t xs, double *x0, double *xfo, double *Uo, int32_t *m, void *s)
{ (void)k; (void)ns; (void)nc; (void)T; (void)nr; (void)kc; (void)o; (void)f; (void)c; (void)b; (void)X; (void)U; (void)xf; (void)xs; (void)x0; (void)xfo; (void)Uo; (void)m; (void)s; UNSUPPORTED("dispatch_gather"); }
int32_t dpilqr_dispatch_gather_params(int32_t c, int32_t kc, int32_t w, int32_t eb, const int32_t *m, const void *src, void *out, void *s)
{ (void)c; (void)kc; (void)w; (void)eb; (void)m; (void)src; (void)out; (void)s; UNSUPPORTED("dispatch_gather_params"); }
int32_t dpilqr_dispatch_stitch(int32_t S, int32_t k, int32_t ns, int32_t nc, int32_t T, const uint64_t *b, const int32_t *rep, const int32_t *size,
                               const int32_t *slot, const dpilqr_bucket_results *R, double *X, double *U, void *s)
{ (void)S; (void)k; (void)ns; (void)nc; (void)T; (void)b; (void)rep; (void)size; (void)slot; (void)R; (void)X; (void)U; (void)s; UNSUPPORTED("dispatch_stitch"); }
int32_t dpilqr_dispatch_pack_rows(int32_t S, int32_t k, int32_t ns, int32_t nc, int32_t T, const uint64_t *b, const int32_t *rep, const int32_t *size,
                                  const int32_t *slot, const dpilqr_bucket_results *R, int32_t *ro, int32_t *nr, double *rows, int64_t rl, void *s)
{ (void)S; (void)k; (void)ns; (void)nc; (void)T; (void)b; (void)rep; (void)size; (void)slot; (void)R; (void)ro; (void)nr; (void)rows; (void)rl; (void)s; UNSUPPORTED("dispatch_pack_rows"); }
int32_t dpilqr_dispatch_scatter_rows(int64_t n, int32_t k, int32_t ns, int32_t nc, int32_t T, const double *rows, int64_t rl, double *X, double *U, void *s)
{ (void)n; (void)k; (void)ns; (void)nc; (void)T; (void)rows; (void)rl; (void)X; (void)U; (void)s; UNSUPPORTED("dispatch_scatter_rows"); }
int32_t dpilqr_random_setup(int32_t S, int64_t seed0, int32_t k, int32_t ns, int32_t nd, double var, double e, double *x0, double *xf, void *s)
{ (void)S; (void)seed0; (void)k; (void)ns; (void)nd; (void)var; (void)e; (void)x0; (void)xf; (void)s; UNSUPPORTED("random_setup"); }
