/*
 * dpilqr_hip.h -- C ABI of libdpilqr_hip.so: the MI355X (gfx950) batched iLQR hot path.
 *
 * This is the drop-in boundary for the hot path of labicon/dp-ilqr.  It replaces
 *   - the Cython FFI  dpilqr/bbdynamicswrap.pyx:61-164  (f / integrate / linearize + Model enum :8-16)
 *     over the static C++ functions of dpilqr/bbdynamics.cpp:39-711, and
 *   - the Python inner loops that call it: ilqrSolver._rollout / _backward_pass / _forward_pass /
 *     solve (dpilqr/control.py:80-225), GameCost.__call__/quadraticize (dpilqr/cost.py:197-239),
 *     and the per-agent sub-problem loop of solve_distributed (dpilqr/distributed.py:55-97).
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types.  Every entry point returns 0 on success or a
 *     negative DPILQR_E* code and never throws or aborts -- on the host or on the device: no kernel of the
 *     library traps; dpilqr_last_error() holds the message of the calling thread's last failure.
 *   - the environment does not steer the library: the A/B route switches of its experiments (DPILQR_NO_FUSED,
 *     DPILQR_MFMA_WAVES, DPILQR_BIG_TEAM, ...) are ignored unless the process sets DPILQR_DEBUG_ROUTES=1.
 *   - every data pointer is a DEVICE pointer owned by the caller (torch.Tensor.data_ptr()); nothing is
 *     allocated behind the caller's back.  Workspace sizes come from *_workspace_bytes.  The one piece of
 *     host-side state -- the pinned mailbox and events through which the synchronous solve follows the device --
 *     is an explicit object, dpilqr_solver_create / dpilqr_solver_destroy.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); 0 = null stream.
 *     Every entry point only ENQUEUES work on `stream` and returns, with one exception: dpilqr_solve_batch[_f32]
 *     returns when the solve has finished (the number of iLQR iterations is data dependent and it stops launching
 *     as soon as the device reports that nothing is left).  dpilqr_solve_enqueue is the same solve as pure enqueue.
 *   - all real data is IEEE fp64, row-major, C-contiguous; integer arrays are int32.  The *_f32 entry points are the
 *     fp32 arm of BASELINE config 5's tolerance study: same semantics, trajectories and gains in float.
 *   - a "batch" is B independent sub-problems of identical shape: k agents x (n_s states, n_c controls),
 *     horizon T.  Joint dims n_x = k*n_s, n_u = k*n_c.  The reference assumes the same uniformity
 *     (dynamics.py:165-166, util.py:97-109).
 */
#ifndef DPILQR_HIP_H
#define DPILQR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPILQR_ABI_VERSION 4   /* 3: t_kill in dpilqr_solve_batch[_f32] / dpilqr_solve_enqueue, DPILQR_STATUS_KILLED
                                  4: DPILQR_STATUS_FAULT; route switches honoured only under DPILQR_DEBUG_ROUTES=1 */

/* error codes */
#define DPILQR_OK 0
#define DPILQR_EINVAL (-1)     /* bad argument (null pointer, size out of range, unknown model)   */
#define DPILQR_EUNSUPPORTED (-2) /* shape outside what the kernels implement (e.g. n_x too large)   */
#define DPILQR_EHIP (-3)       /* a HIP runtime call failed, or a kernel gave an item up (STATUS_FAULT)  */
#define DPILQR_ENOGPU (-4)     /* no usable gfx950 device                                         */
#define DPILQR_EWORKSPACE (-5) /* workspace too small                                             */

/* Model enum -- identical values to bbdynamicswrap.pyx:8-16 */
#define DPILQR_MODEL_DOUBLE_INT_4D 0
#define DPILQR_MODEL_DOUBLE_INT_6D 1
#define DPILQR_MODEL_CAR_3D 2
#define DPILQR_MODEL_UNICYCLE_4D 3
#define DPILQR_MODEL_QUADCOPTER_6D 4
#define DPILQR_MODEL_HUMAN_6D 5
#define DPILQR_MODEL_HUMAN_LIN_6D 6
#define DPILQR_MODEL_QUADCOPTER_12D 7
/* Not a reference model.  BASELINE config 5 mixes QuadcopterDynamics12D (12 states / 4 controls) with HumanDynamics6D
 * (6 / 3) "zero-padded": the reference cannot stack agents of different dims (dynamics.py:165-170, util.py:229-236), so
 * this library defines the padded human itself -- HumanDynamics6D (bbdynamics.cpp:308-391) in the first 6 states and 3
 * controls, six padded states that never move (A = 1 on their diagonal, B = 0) and a fourth control that does nothing
 * (give it a small positive R entry, as scripts/examples.py:93 does for the human's unused third control). */
#define DPILQR_MODEL_HUMAN_PAD_12D 8

/* per-item solve status (written by dpilqr_solve_batch) */
#define DPILQR_STATUS_ACTIVE 0
#define DPILQR_STATUS_CONVERGED 1        /* |dJ/J*| < tol on an accepted step, control.py:184,210 */
#define DPILQR_STATUS_LINESEARCH_FAILED 2 /* all alphas rejected, control.py:195-198             */
#define DPILQR_STATUS_MAX_ITER 3          /* n_lqr_iter exhausted                                 */
#define DPILQR_STATUS_SINGULAR 4          /* exactly zero pivot in Q_uu (np.linalg.solve would raise) */
#define DPILQR_STATUS_KILLED 5            /* t_kill: the item's own solve time ran out after an accepted, not yet
                                             converged iteration, control.py:213-218; X, U = that accepted iterate   */
#define DPILQR_STATUS_FAULT 6             /* the device gave this item up: a hand-over inside a team of workgroups (n_x > 60 with
                                             fewer items than CUs) did not arrive within its bound.  No gains exist for the
                                             iteration; X, U = the last accepted iterate.  dpilqr_solve_batch[_f32] then returns
                                             DPILQR_EHIP with the count in dpilqr_last_error(); the other items' results are
                                             valid and the HIP context stays usable.  Should not occur (a helper that joined a
                                             team is resident); it replaces what used to be a device-side abort.          */

#define DPILQR_N_ALPHA 10 /* ilqrSolver.N_LS_ITER, control.py:51 */

/*
 * Batch descriptor: the device-side lowering of B ilqrProblem objects (problem.py:15-25) whose
 * dynamics are MultiDynamicalModel([CppModel...]) (dynamics.py:133-157) and whose cost is
 * GameCost([ReferenceCost...], ProximityCost) (cost.py:174-195).
 * `*_bstride` is the element stride between consecutive batch items; 0 shares one copy.
 */
typedef struct dpilqr_batch_desc {
    int32_t B;   /* sub-problems in the batch                                    */
    int32_t k;   /* agents per sub-problem                                       */
    int32_t n_s; /* per-agent state dim   (3, 4, 6 or 12)                        */
    int32_t n_c; /* per-agent control dim (2, 3 or 4)                            */
    int32_t T;   /* horizon N (control.py:56)                                    */
    int32_t uniform_model; /* hints, 0 = unknown / mixed (always valid).  Bits 0..7: 1 + Model enum when EVERY agent of
                              EVERY item uses that model; bits 8..15: 1 + n_dims when every agent of every item has
                              that ProximityCost.n_dims; bit 16: every agent has the same Q, the same R and the same Q_f
                              (and their batch strides are 0); bit 17: every agent of every item is a DoubleIntDynamics4D
                              or a UnicycleDynamics4D (models may be mixed).  They let the solver pick kernels compiled
                              per model and skip work that cannot depend on the item.                           */
    double dt;     /* DynamicalModel.dt                                          */
    double w_ref;  /* GameCost.REF_WEIGHT  = 1   (cost.py:185)                   */
    double w_prox; /* GameCost.PROX_WEIGHT = 200 (cost.py:186)                   */
    const int32_t* model;  int64_t model_bstride;  /* [B][k] Model enum per agent                 */
    const int32_t* n_dims; int64_t n_dims_bstride; /* [B][k] ProximityCost.n_dims (cost.py:111)   */
    const double* xf;      int64_t xf_bstride;     /* [B][k*n_s] ReferenceCost.xf                 */
    const double* Q;       int64_t Q_bstride;      /* [B][k][n_s*n_s]                             */
    const double* R;       int64_t R_bstride;      /* [B][k][n_c*n_c]                             */
    const double* Qf;      int64_t Qf_bstride;     /* [B][k][n_s*n_s]                             */
    const double* radius;  int64_t radius_bstride; /* [B] ProximityCost.radius                    */
} dpilqr_batch_desc;

/* ------------------------------------------------------------------ library */
int32_t dpilqr_abi_version(void);
const char* dpilqr_last_error(void);
/* 0 if device `dev` exists and is gfx950; fills (may be NULL) its CU count and LDS bytes per workgroup */
int32_t dpilqr_device_info(int32_t dev, int32_t* n_cu, int32_t* lds_bytes, char* arch, int32_t arch_len);
int32_t dpilqr_model_dims(int32_t model, int32_t* n_s, int32_t* n_c);

/* -------------------------------------------------- (1) model FFI, batched over n agents
 * replaces bbdynamicswrap.f / integrate / linearize (pyx:61-164).  model[n]; x[n][n_s]; u[n][n_c]
 * with n_s/n_c taken from `family_ns` (all n agents must belong to models of that state dim).     */
int32_t dpilqr_model_f(int32_t n, int32_t family_ns, const int32_t* model, const double* x, const double* u,
                       double* x_dot, void* stream);
int32_t dpilqr_model_integrate(int32_t n, int32_t family_ns, const int32_t* model, const double* x,
                               const double* u, double dt, double* x_new, void* stream);
int32_t dpilqr_model_linearize(int32_t n, int32_t family_ns, const int32_t* model, const double* x,
                               const double* u, double dt, double* A /*[n][n_s*n_s]*/,
                               double* B /*[n][n_s*n_c]*/, void* stream);

/* -------------------------------------------------- (2) joint cost, batched over B x n_pts points
 * replaces GameCost.__call__ / quadraticize (cost.py:197-239).  x[B][n_pts][n_x], u[B][n_pts][n_u].  */
int32_t dpilqr_cost_eval(const dpilqr_batch_desc* desc, int32_t n_pts, const double* x, const double* u,
                         int32_t terminal, double* cost /*[B][n_pts]*/, void* stream);

/* -------------------------------------------------- (3) tiles: the plugin contract of the sweep
 * One "tile record" per (item, time step) holds what DynamicalModel.linearize (dynamics.py:173-186)
 * and Cost.quadraticize (cost.py:208-239) return at (X[t],U[t]), packed as
 *     [ AB n_x*(n_x+n_u) | L_xx n_x*n_x | UG n_u*(n_u+n_x) | L_x n_x | L_u n_u ]
 * where AB stores A and B interleaved by row, row l = [ A[l][0..n_x) | B[l][0..n_u) ], and UG stores L_uu and
 * L_ux the same way, row a = [ L_uu[a][0..n_u) | L_ux[a][0..n_x) ] (the sweep consumes both as stacked matrices).
 * Record t = T holds the terminal quadraticisation (only L_xx, L_x are read).
 * dpilqr_tile_layout returns, for the 7 components in the order A,B,Lxx,Lux,Luu,Lx,Lu, the offset of
 * element [0][0] and the row stride (both in doubles), plus the record stride; a tile buffer is
 * [B][T+1][stride] doubles.  Component offsets and the record stride are 16-byte aligned. */
int32_t dpilqr_tile_layout(int32_t n_x, int32_t n_u, int64_t offsets[7], int64_t row_strides[7], int64_t* stride);
/* device-side producer for the recognised plugin types; X[B][T+1][n_x], U[B][T][n_u].
 * items/n_items select a subset (device int32 list + device count); NULL/NULL = all B items.  Record
 * r of the output belongs to the r-th LISTED item.                                                   */
int32_t dpilqr_make_tiles(const dpilqr_batch_desc* desc, const double* X, const double* U, double* tiles,
                          const int32_t* items, const int32_t* n_items, void* stream);

/* -------------------------------------------------- (4) passes
 * ilqrSolver._rollout (control.py:80-93): X[B][T+1][n_x] (X[:,0] is written from x0), J[B].        */
int32_t dpilqr_rollout(const dpilqr_batch_desc* desc, const double* x0, const double* U, double* X, double* J,
                       void* stream);
/* ilqrSolver._backward_pass (control.py:116-148) on tile records: the Riccati sweep.
 * mu[B] per-item regularisation; K[B][T][n_u][n_x]; d[B][T][n_u]; singular[B] (may be NULL) set to 1
 * where a pivot was exactly zero.  With an item list (items/n_items as above) the tile records, K and
 * d are indexed by POSITION in the list (mu, singular by item id); without one the two coincide.     */
int32_t dpilqr_backward_pass_tiles(int32_t B, int32_t T, int32_t n_x, int32_t n_u, const double* tiles,
                                   const double* mu, double* K, double* d, int32_t* singular,
                                   const int32_t* items, const int32_t* n_items, void* stream);
/* The same sweep for records whose A and B are block diagonal with n_x/block_ns agents' blocks of
 * block_ns x block_ns and block_ns x block_nc -- what MultiDynamicalModel.linearize (dynamics.py:173-186,
 * uniform_block_diag) always returns.  The records are the same dense records and are read in full; the
 * caller's promise lets the sweep skip the products with the structural zeros (which are exact, so the
 * gains are those of the dense sweep).  block_ns == 0 is dpilqr_backward_pass_tiles.                   */
int32_t dpilqr_backward_pass_tiles_blocks(int32_t B, int32_t T, int32_t n_x, int32_t n_u, int32_t block_ns,
                                          int32_t block_nc, const double* tiles, const double* mu, double* K,
                                          double* d, int32_t* singular, const int32_t* items,
                                          const int32_t* n_items, void* stream);
/* convenience: make_tiles + backward_pass_tiles for recognised plugins; workspace = tile buffer
 * (dpilqr_tiles_bytes always suffices; dpilqr_backward_pass_workspace_bytes(desc, 8) is exact).  Clusters with
 * n_x > 60 take the fused sweep of the large-cluster path: it evaluates linearize / quadraticize per step inside the
 * sweep and uses the workspace as its scratch, no tile records exist there (a record would be 1.28 MB at n_x = 240). */
int64_t dpilqr_tiles_bytes(int32_t B, int32_t T, int32_t n_x, int32_t n_u);
int32_t dpilqr_backward_pass(const dpilqr_batch_desc* desc, const double* X, const double* U, const double* mu,
                             double* K, double* d, double* tiles_workspace, void* stream);
/* The same backward pass without tile records ("fused variant"): the sweep evaluates linearize / quadraticize
 * itself and reads only (X, U).  Gains are bit-identical to dpilqr_backward_pass.  Served: (a) batches whose descriptor hints
 * say "DoubleIntDynamics4D agents only (at most five), n_dims = 2 everywhere, one Q, R, Q_f for every agent of every item"
 * (12 KB instead of 535 KB per cfg2 pass); (a') at most five agents of the planar four-state models -- DoubleIntDynamics4D
 * and UnicycleDynamics4D, mixed or not (the descriptor's model hint or its bit 17) -- with n_dims = 2 everywhere and ANY
 * per-agent, per-item Q, R, Q_f;
 * (b) 6..15 agents of the four-state family or 5..10 of the six-state family, any
 * models of the family, any weights (cfg3 / cfg4 clusters: 90 doubles instead of a 94 KB record per step at n_x = 60);
 * (c) 1..4 agents of the six-state family or 1..6 CarDynamics3D agents, any models of the family, any weights, any n_dims
 * (the plugins evaluated inside the padded wavefront sweep).
 * DPILQR_EUNSUPPORTED for any other batch (twelve-state agents, four-state clusters of at most five agents without
 * the hints).  dpilqr_solve_batch picks it by itself, and its workspace then holds no records. */
int32_t dpilqr_backward_pass_fused(const dpilqr_batch_desc* desc, const double* X, const double* U, const double* mu, double* K,
                                   double* d, int32_t* singular, void* stream);
/* ilqrSolver._forward_pass (control.py:95-114) for n_alpha step sizes at once:
 * Xn[B][n_alpha][T+1][n_x], Un[B][n_alpha][T][n_u], Jn[B][n_alpha].                                  */
int32_t dpilqr_forward_pass(const dpilqr_batch_desc* desc, const double* X, const double* U, const double* K,
                            const double* d, const double* alphas, int32_t n_alpha, double* Xn, double* Un,
                            double* Jn, void* stream);
/* the float32-rounded line-search table of control.py:162 (host array of DPILQR_N_ALPHA doubles) */
int32_t dpilqr_alphas(double* alphas_host);

/* -------------------------------------------------- (5) the whole solve, device resident
 * ilqrSolver.solve (control.py:150-225) for every item of the batch; one launch sequence per iLQR
 * iteration over the items still active (finished items are retired through a device-side list).
 *   x0[B][n_x]; U[B][T][n_u] in: warm start, out: solution; X[B][T+1][n_x] out;
 *   J[B] out = last EVALUATED forward-pass cost (reference quirk, control.py:181,225);
 *   status[B], n_bwd[B], n_fwd[B] out; trace (may be NULL) [B][n_lqr_iter][5] =
 *   (mu_before, accepted alpha index or -1, J_last, J_star_after, n_forward_passes).
 *   K_out/d_out (may be NULL): [B][T][n_u][n_x] / [B][T][n_u] gains of each item's LAST backward pass.
 * Plugins the library does not recognise never reach this entry point: the host solver calls their
 * linearize/quadraticize itself and feeds dpilqr_backward_pass_tiles.
 * window (<= 0: B): the most sub-problems in flight at once.  Sub-problems need very different numbers
 *   of iterations (1..25 at cfg2), so finished ones are retired on the device and their places refilled
 *   from the not-yet-started items of the batch: every launch stays near `window` items, and the large
 *   per-iteration buffers (tile records, gains, line-search candidates) are sized by `window`, not B.
 *   Hand the solver the whole Monte-Carlo batch and let `window` bound the memory.
 * t_kill (seconds; <= 0 or NaN: none): the reference's real-time bail-out (control.py:213-218; scripts/analysis.py:145-147
 *   runs its default study with t_kill = dt).  There the check sits at the end of an iteration whose step was accepted
 *   and did not converge: `perf_counter() - t0 > t_kill` -> break, t0 taken after the initial rollout.  Here every item has
 *   its own t0 -- the device's constant-rate clock (s_memrealtime) at the moment the item is ADMITTED to the window, i.e.
 *   just before its first backward pass -- and the line-search kernel that takes the item's accept / converge decision
 *   reads the same clock: elapsed > t_kill -> status DPILQR_STATUS_KILLED, the item is retired with the iterate it has
 *   just accepted (so every item runs at least one iteration, as in the reference, and J/X/U are exactly what a solve
 *   with n_lqr_iter = n_bwd[item] returns).  The decision is taken on the device: no host read, no launch-ahead lag,
 *   and dpilqr_solve_enqueue honours it as well.
 *   CONSEQUENCE of a shared device: the item's elapsed time is device wall time from its admission, and every launch it
 *   takes part in also carries up to `window` other items -- the clock includes their work.  Which items are killed
 *   therefore depends on B, `window` and whatever else the GPU runs, and differs from run to run; it is not the set the
 *   reference would kill (one solve alone on a CPU core), and studies run with t_kill cannot be compared row by row with
 *   the reference's.  n_lqr_iter is the deterministic budget (a KILLED item equals the n_lqr_iter = n_bwd solve).
 * workspace: dpilqr_solve_workspace_bytes(desc, window, K_out == NULL) bytes of device memory.           */
int64_t dpilqr_solve_workspace_bytes(const dpilqr_batch_desc* desc, int32_t window, int32_t gains_in_workspace);
/* Host-side state of the synchronous solve: a pinned mailbox (a few words the device posts its active-list counters
 * into), the events that tell the host when a post has landed, and the optional profiler.  Bound to the device that is
 * current at creation; one solver serves one solve at a time (use one per host thread / stream).  solver == NULL in
 * dpilqr_solve_batch selects a per-thread default that is created on first use -- a convenience for scripts; a caller
 * that must not see an allocation inside a solve creates its solver up front. */
typedef struct dpilqr_solver dpilqr_solver;
int32_t dpilqr_solver_create(dpilqr_solver** out);
int32_t dpilqr_solver_destroy(dpilqr_solver* solver);
/* Progress of a synchronous solve, for callers that move finished results while the rest still solves (the multi-GPU
 * form overlaps the path's one collective -- the all-gather of converged trajectories, SURVEY 8(e) -- with the solve:
 * dpilqr_amd/sharding.py).  Items are admitted in index order; `n_finished` is a PREFIX: X, U, status, n_bwd, n_fwd of
 * items [0, n_finished) are final in device memory when the callback runs (J is written when the solve ends), so work
 * on them may be enqueued on ANOTHER stream at once (the solve's own stream is still busy).  Called from inside dpilqr_solve_batch on the calling thread,
 * with non-decreasing n_finished (a few iterations late: the host follows the device's counters), and a last time with
 * n_finished = n_items after the solve's stream has been waited for.  fn = NULL clears.  solver = NULL: the calling
 * thread's default solver.  (Replaces nothing in the reference: its pool hands results back one by one,
 * distributed.py:86-93.) */
typedef void (*dpilqr_progress_fn)(void* user, int32_t n_finished, int32_t n_items);
int32_t dpilqr_solver_set_progress(dpilqr_solver* solver, dpilqr_progress_fn fn, void* user);
/* Synchronous, adaptive: launches iterations until the device reports that every item has finished (it follows the
 * device's counters a few iterations late, so launches are always queued ahead), then waits for `stream`. */
int32_t dpilqr_solve_batch(dpilqr_solver* solver, const dpilqr_batch_desc* desc, const double* x0, double* U,
                           int32_t n_lqr_iter, double tol, double t_kill, int32_t window, void* workspace, int64_t workspace_bytes,
                           double* X, double* J, int32_t* status, int32_t* n_bwd, int32_t* n_fwd, double* trace,
                           double* K_out, double* d_out, void* stream);
/* Enqueue-only: no host-side state, no allocation, no host read, no synchronisation -- it can be queued behind other
 * work, overlapped with another solve on another stream, or captured in a hipGraph.  Exactly n_global_iter global
 * iterations (backward pass + line search over the active list) are launched with window-wide grids; launches that
 * find the active list empty exit at once.  Whether everything finished is for the caller to read from `status`
 * (DPILQR_STATUS_ACTIVE = not yet); dpilqr_solve_iterations_bound gives the n_global_iter that always suffices, a
 * smaller number plus a second call with resume = 1 (same arguments, same workspace) continues where the first
 * stopped.  resume = 0 initialises the solver state and rolls out (x0, U); J, X, U, status ... hold the state reached. */
int32_t dpilqr_solve_enqueue(const dpilqr_batch_desc* desc, const double* x0, double* U, int32_t n_lqr_iter, double tol,
                             double t_kill, int32_t window, void* workspace, int64_t workspace_bytes, double* X, double* J,
                             int32_t* status, int32_t* n_bwd, int32_t* n_fwd, double* trace, double* K_out,
                             double* d_out, int32_t n_global_iter, int32_t resume, void* stream);
int64_t dpilqr_solve_iterations_bound(const dpilqr_batch_desc* desc, int32_t window, int32_t n_lqr_iter);

/* -------------------------------------------------- (5b) the fp32 arm (BASELINE config 5: fp32 vs fp64 tolerance study)
 * Same passes and the same solve with trajectories, gains and every intermediate in float (the descriptor stays fp64
 * and is rounded on use; costs J, the regularisation state and the decision trace are reported in double).  They run
 * on the size-generic kernels of the large-cluster path for every shape. */
int32_t dpilqr_rollout_f32(const dpilqr_batch_desc* desc, const float* x0, const float* U, float* X, double* J, void* stream);
int64_t dpilqr_backward_pass_workspace_bytes(const dpilqr_batch_desc* desc, int32_t elem_bytes /* 8: dpilqr_backward_pass, 4: _f32 */);
int32_t dpilqr_backward_pass_f32(const dpilqr_batch_desc* desc, const float* X, const float* U, const double* mu, float* K,
                                 float* d, void* workspace, void* stream);
int32_t dpilqr_forward_pass_f32(const dpilqr_batch_desc* desc, const float* X, const float* U, const float* K, const float* d,
                                const double* alphas, int32_t n_alpha, float* Xn, float* Un, double* Jn, void* stream);
int64_t dpilqr_solve_workspace_bytes_f32(const dpilqr_batch_desc* desc, int32_t window, int32_t gains_in_workspace);
int32_t dpilqr_solve_batch_f32(dpilqr_solver* solver, const dpilqr_batch_desc* desc, const float* x0, float* U,
                               int32_t n_lqr_iter, double tol, double t_kill, int32_t window, void* workspace, int64_t workspace_bytes,
                               float* X, double* J, int32_t* status, int32_t* n_bwd, int32_t* n_fwd, double* trace,
                               float* K_out, float* d_out, void* stream);

/* -------------------------------------------------- measurement hooks (bench.py's roofline leg)
 * When enabled, dpilqr_solve_batch brackets every kernel launch of its iteration loop with HIP events
 * recorded on `stream` and accumulates, in the solver object it was given, for each kernel class c
 * (0 = tile producer, 1 = Riccati sweep, 2 = line search / forward pass, 3 = initial rollout):
 * total milliseconds, number of launches, and number of sub-problems those launches processed.
 * enable: bit 0 = on; bits 1..4 = optional mask of the classes to bracket (0 = all).  Every bracketed launch
 * costs a dispatch gap, so a measurement of one kernel asks for that class only.                       */
int32_t dpilqr_profile_enable(dpilqr_solver* solver /* NULL: the calling thread's default solver */, int32_t enable);
/* diagnostic: register (or clear with NULL) a device buffer of 4 x uint64 per sweep workgroup that receives
 * {start, end} wall-clock stamps (100 MHz) and the HW_ID / XCC_ID registers of the wave that ran it.   */
int32_t dpilqr_debug_stamps(void* device_buffer);
int32_t dpilqr_profile_read(dpilqr_solver* solver, double ms[4], int64_t launches[4], int64_t items[4], int32_t reset);
/* the wavefront sweep's share of class 1, by variant: waves = wavefronts per workgroup (4, 8 or 12), i.e. the
 * k_riccati_mfma<n_x, n_u, waves, ...> instantiation a rocprofv3 kernel trace lists under that name               */
int32_t dpilqr_profile_read_sweep(dpilqr_solver* solver, int32_t waves, double* ms, int64_t* launches, int64_t* items, int32_t reset);

/* -------------------------------------------------- (6) dispatch front end ("next" row)
 * define_inter_graph_threshold (distributed.py:224-247) for S scenarios at once:
 * X[S][N][k*n_s] sampled trajectories (N may be 1), adjacency out adj[S][k][k] (int32, incl. self). */
int32_t dpilqr_pairwise_graph(int32_t S, int32_t N, int32_t k, int32_t n_s, const double* X,
                              const double* radius /*[S]*/, int32_t* adj, void* stream);

/* -------------------------------------------------- (7) dispatch front and back end for S scenarios ("next" row f1)
 * Everything solve_distributed does around the solves (distributed.py:42-77,100-101) for S Monte-Carlo scenarios of ONE
 * k-agent problem, as device array work over (S, k); k <= DPILQR_MAX_AGENTS.  The reference solves one sub-problem per
 * AGENT (its closed neighbourhood) and agents with the same neighbourhood repeat the same solve (quirk Q11): here every
 * distinct (scenario, neighbourhood) is solved once and its owners' columns are copied out.
 *   dispatch_graph   X[S][N][k*n_s] (N = 1: the scenario's state; N = T+1: a trajectory, sampled like :229-235), radius[S]
 *                    (stride 0: shared) -> bits[S*k] neighbourhood masks (bit j = agent j, incl. itself), rep[S*k] the first
 *                    agent with the same mask (-1 for ignored agents: ignore[k] may be NULL), size[S*k] agents in the mask,
 *                    and a stable sort of the representatives by size: order[pos] = s*k + i, bucket_start / bucket_count[k+1]
 *                    (index = cluster size), slot[S*k] = position of a representative inside its bucket (-1 otherwise).
 *   dispatch_gather  inputs of the sub-problems [first, first+count) of the size-kc bucket: x0, x_f [count][kc*n_s],
 *                    U0 [count][T][kc*n_c], members [count][kc] (NULL if not needed), from X (row 0), U, xf of their scenarios.
 *   dispatch_gather_params  per-agent parameters of a heterogeneous team -> per-item arrays: out[j][p][w] = src[members[j][p]][w].
 *   dispatch_stitch  the owners' columns of the solved sub-problems -> X_dec[S][T+1][k*n_s], U_dec[S][T][k*n_c] (ignored agents
 *                    keep what the buffers held; callers zero them).  results: per cluster size the solved X [count][T+1][kc*n_s],
 *                    U [count][T][kc*n_c] and which slice [first, first+count) of the bucket they are.
 *   dispatch_pack_rows / dispatch_scatter_rows  the multi-GPU form: a rank that solved a slice of every bucket packs one row
 *                    [s*k+i | X columns | U columns] per (scenario, agent) it owns (row_of[S*k], n_rows out; rows == NULL only
 *                    counts); after the path's one all-gather every rank scatters all ranks' rows (index < 0: padding). */
#define DPILQR_MAX_AGENTS 64
typedef struct dpilqr_bucket_results {
    const double* X[DPILQR_MAX_AGENTS + 1];
    const double* U[DPILQR_MAX_AGENTS + 1];
    int32_t first[DPILQR_MAX_AGENTS + 1];
    int32_t count[DPILQR_MAX_AGENTS + 1];
} dpilqr_bucket_results;
int32_t dpilqr_dispatch_graph(int32_t S, int32_t N, int32_t k, int32_t n_s, const double* X, const double* radius,
                              int64_t radius_stride, const int32_t* ignore, uint64_t* bits, int32_t* rep, int32_t* size,
                              int32_t* order, int32_t* slot, int32_t* bucket_start, int32_t* bucket_count, void* stream);
int32_t dpilqr_dispatch_gather(int32_t k, int32_t n_s, int32_t n_c, int32_t T, int32_t n_rows, int32_t kc, const int32_t* order,
                               int32_t first, int32_t count, const uint64_t* bits, const double* X, const double* U,
                               const double* xf, int64_t xf_stride, double* x0_out, double* xf_out, double* U_out,
                               int32_t* members, void* stream);
int32_t dpilqr_dispatch_gather_params(int32_t count, int32_t kc, int32_t width, int32_t elem_bytes, const int32_t* members,
                                      const void* src, void* out, void* stream);
int32_t dpilqr_dispatch_stitch(int32_t S, int32_t k, int32_t n_s, int32_t n_c, int32_t T, const uint64_t* bits, const int32_t* rep,
                               const int32_t* size, const int32_t* slot, const dpilqr_bucket_results* results, double* X_dec,
                               double* U_dec, void* stream);
int32_t dpilqr_dispatch_pack_rows(int32_t S, int32_t k, int32_t n_s, int32_t n_c, int32_t T, const uint64_t* bits,
                                  const int32_t* rep, const int32_t* size, const int32_t* slot,
                                  const dpilqr_bucket_results* results, int32_t* row_of, int32_t* n_rows, double* rows,
                                  int64_t row_len, void* stream);
int32_t dpilqr_dispatch_scatter_rows(int64_t n_rows_total, int32_t k, int32_t n_s, int32_t n_c, int32_t T, const double* rows,
                                     int64_t row_len, double* X_dec, double* U_dec, void* stream);

/* -------------------------------------------------- (8) scenario generation on the device ("next" row f4)
 * x0, x_f [S][k*n_s] of the scenarios s = seed0 .. seed0 + S - 1 of scripts/analysis.py:45-54: what
 *     np.random.seed(s); random_setup(k, n_s, is_rotation=False, rel_dist=., var=var, n_d=n_d, random=True, energy=energy)
 * returns (util.py:125-217), bit for bit -- NumPy's legacy MT19937 stream, its uniform(), its mean / norm / pairwise-sum
 * orders -- so that a Monte-Carlo driver needs neither a host loop over seeds nor an upload.  energy = 0: no normalisation. */
int32_t dpilqr_random_setup(int32_t S, int64_t seed0, int32_t k, int32_t n_s, int32_t n_d, double var, double energy,
                            double* x0, double* xf, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DPILQR_HIP_H */
