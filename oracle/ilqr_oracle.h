/*
 * ilqr_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, scalar, fp64 restatement of the reference's iLQR hot path
 * (labicon/dp-ilqr: dpilqr/control.py, cost.py, dynamics.py, bbdynamics.cpp).
 * It exists so that the HIP kernels can be checked against an independent
 * CPU implementation on the GPU box, where the reference itself cannot run.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this library.  The product (dpilqr_amd/) never links, imports or
 * executes it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against the .npz files under tests/golden/, which were produced by running the real
 * reference in the build container (tests/golden/make_golden.py).
 */
#ifndef ILQR_ORACLE_H
#define ILQR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Model enum values: bbdynamicswrap.pyx:8-16 (declaration order, 0..7); 8 is this build's padded human. */
enum {
    ORACLE_DOUBLE_INT_4D = 0,
    ORACLE_DOUBLE_INT_6D = 1,
    ORACLE_CAR_3D = 2,
    ORACLE_UNICYCLE_4D = 3,
    ORACLE_QUADCOPTER_6D = 4,
    ORACLE_HUMAN_6D = 5,
    ORACLE_HUMAN_LIN_6D = 6,
    ORACLE_QUADCOPTER_12D = 7,
    ORACLE_HUMAN_PAD_12D = 8, /* not a reference model: HumanDynamics6D zero-padded to 12 states / 4 controls (cfg5) */
    ORACLE_N_MODELS = 9
};

/* One centralised (sub)problem: k agents with uniform per-agent dims
 * (dynamics.py:165-166 slices every agent with x_dims[0]). */
typedef struct {
    int k;               /* number of agents                               */
    int n_s, n_c;        /* per-agent state / control dimension            */
    int T;               /* horizon N (control.py:56)                      */
    double dt;
    const int *model;    /* [k]   Model enum per agent                     */
    const int *n_dims;   /* [k]   position dims per agent (cost.py:111)    */
    const double *xf;    /* [k*n_s] goal state                             */
    const double *Q;     /* [k][n_s*n_s] row-major                         */
    const double *R;     /* [k][n_c*n_c]                                   */
    const double *Qf;    /* [k][n_s*n_s]                                   */
    double radius;
    double w_ref;        /* GameCost.REF_WEIGHT  (cost.py:185)             */
    double w_prox;       /* GameCost.PROX_WEIGHT (cost.py:186)             */
} oracle_problem;

int oracle_model_dims(int model, int *n_s, int *n_c);
int oracle_model_f(int model, const double *x, const double *u, double *xdot);
int oracle_model_integrate(int model, const double *x, const double *u, double dt, double *xn);
int oracle_model_linearize(int model, const double *x, const double *u, double dt, double *A, double *B);

/* cost.py:269-315; g[3], H[9] (3x3 row-major), entries beyond n_d are zero */
void oracle_quadraticize_distance(const double *pa, const double *pb, double radius, int n_d,
                                  double *g, double *H);

double oracle_cost(const oracle_problem *p, const double *x, const double *u, int terminal);
void oracle_quadraticize(const oracle_problem *p, const double *x, const double *u, int terminal,
                         double *Lx, double *Lu, double *Lxx, double *Luu, double *Lux);
double oracle_prox_cost(const oracle_problem *p, const double *x);
void oracle_prox_quadraticize(const oracle_problem *p, const double *x, double *Lx, double *Lxx);
void oracle_step(const oracle_problem *p, const double *x, const double *u, double *xn);
void oracle_linearize(const oracle_problem *p, const double *x, const double *u, double *A, double *B);

/* control.py:80-93 */
double oracle_rollout(const oracle_problem *p, const double *x0, const double *U, double *X);
/* control.py:116-148 ; returns 0, or -1 if a pivot was exactly zero */
int oracle_backward_pass(const oracle_problem *p, const double *X, const double *U, double mu,
                         double *K, double *d);
/* the same recursion fed with explicit per-step tiles (the plugin contract) */
int oracle_backward_pass_tiles(int n_x, int n_u, int T, const double *A, const double *B,
                               const double *Lx, const double *Lu, const double *Lxx,
                               const double *Luu, const double *Lux, double mu, double *K, double *d);
/* control.py:95-114 */
double oracle_forward_pass(const oracle_problem *p, const double *X, const double *U, const double *K,
                           const double *d, double alpha, double *Xn, double *Un);

/* float32-rounded line-search table, control.py:162 (quirk Q1) */
void oracle_alphas(double *alphas10);

/* status codes written by oracle_solve */
enum { ORACLE_CONVERGED = 1, ORACLE_LINESEARCH_FAILED = 2, ORACLE_MAX_ITER = 3 };

/* control.py:150-225.  X,U: in U0 / out solution.  trace (may be NULL) is
 * [n_lqr_iter][5] = (mu_before, accepted alpha index or -1, J_last_evaluated,
 * J_star_after, n_forward_passes).  Returns the status code. */
int oracle_solve(const oracle_problem *p, const double *x0, double *U, int n_lqr_iter, double tol,
                 double *X, double *J_out, double *trace, int *n_bwd, int *n_fwd);

/* Batch of B independent problems sharing dims/model/weights (cpu_baseline leg
 * of bench.py).  Arrays are [B][...] contiguous; Q,R,Qf,model,n_dims shared.
 * n_threads <= 0 -> OpenMP default. */
int oracle_solve_batch(const oracle_problem *proto, int B, const double *x0, const double *xf,
                       double *U, int n_lqr_iter, double tol, double *X, double *J,
                       int *status, int *n_bwd, int *n_fwd, int n_threads);
int oracle_solve_batch_trace(const oracle_problem *proto, int B, const double *x0, const double *xf, double *U,
                             int n_lqr_iter, double tol, double *X, double *J, int *status, int *n_bwd,
                             int *n_fwd, int n_threads, double *trace /* [B][max(n_lqr_iter,1)][5] or NULL */);
/* oracle_solve made to FOLLOW the decision trace of an implementation under test (see ilqr_oracle.c): the oracle's numbers
 * for the same iterates, its own verdicts and the distance from equality of every comparison that went the other way.
 * rtrace [n_forced][8] = (mu_before, own accepted index, J_last, J_star_after, accept margin, convergence margin,
 * J_star_before, own converged flag). */
int oracle_solve_replay(const oracle_problem *p, const double *x0, double *U, int n_lqr_iter, double tol,
                        const double *forced, int n_forced, int forced_status, double *X, double *J_out,
                        double *rtrace);
int oracle_replay_batch(const oracle_problem *proto, int B, const double *x0, const double *xf, double *U,
                        int n_lqr_iter, double tol, const double *forced, const int *n_forced,
                        const int *forced_status, double *X, double *J, int *status, double *rtrace, int n_threads);
int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
