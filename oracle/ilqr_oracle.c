/*
 * ilqr_oracle.c -- TEST INFRASTRUCTURE ONLY (see ilqr_oracle.h).
 *
 * Scalar fp64 restatement of the reference hot path, written from the maths in
 * SURVEY.md Appendix A.  Every function names the reference lines it follows.
 * Evaluation order follows the reference where it is defined (NumPy `@` chains
 * associate left to right, costs accumulate in time order, pairs in
 * itertools.combinations order) so that this code tracks the reference to
 * ~1e-13; bit-exactness with BLAS/LAPACK is not a goal.
 */
#include "ilqr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GRAV 9.80665 /* bbdynamics.cpp:11 */

/* Quadcopter12D inertia constants, exact rationals of bbdynamics.cpp:507-510 */
#define Q12_FZ (2000.0 / 63.0)
#define Q12_TX (625000000000000000.0 / 10982593196059.0)
#define Q12_TY (5000000000000000000.0 / 92848985528431.0)
#define Q12_TZ (10000000000000000000.0 / 271597947137541.0)
#define Q12_CX (85899976080679.0 / 175721491136944.0)
#define Q12_CY (95876456000597.0 / 185697971056862.0)
#define Q12_CZ (9976479919918.0 / 271597947137541.0)

#define MAX_NS 12

/* ------------------------------------------------------------------ models */

static const int MODEL_NS[ORACLE_N_MODELS] = {4, 6, 3, 4, 6, 6, 6, 12, 12}; /* dynamics.py:205-250; 8: padded human */
static const int MODEL_NC[ORACLE_N_MODELS] = {2, 3, 2, 2, 3, 3, 3, 4, 4};

int oracle_model_dims(int model, int *n_s, int *n_c)
{
    if (model < 0 || model >= ORACLE_N_MODELS) return -1;
    *n_s = MODEL_NS[model];
    *n_c = MODEL_NC[model];
    return 0;
}

/* continuous dynamics, bbdynamics.cpp:108-117,150-161,230-239,264-274,308-329,393-406,417-429,493-511 */
int oracle_model_f(int model, const double *x, const double *u, double *o)
{
    switch (model) {
    case ORACLE_DOUBLE_INT_4D:
        o[0] = x[2]; o[1] = x[3]; o[2] = u[0]; o[3] = u[1];
        return 0;
    case ORACLE_DOUBLE_INT_6D:
        o[0] = x[3]; o[1] = x[4]; o[2] = x[5]; o[3] = u[0]; o[4] = u[1]; o[5] = u[2];
        return 0;
    case ORACLE_CAR_3D:
        o[0] = u[0] * cos(x[2]); o[1] = u[0] * sin(x[2]); o[2] = u[1];
        return 0;
    case ORACLE_UNICYCLE_4D:
        o[0] = x[2] * cos(x[3]); o[1] = x[2] * sin(x[3]); o[2] = u[0]; o[3] = u[1];
        return 0;
    case ORACLE_QUADCOPTER_6D:
        o[0] = x[3]; o[1] = x[4]; o[2] = x[5];
        o[3] = GRAV * tan(u[2]); o[4] = -GRAV * tan(u[1]); o[5] = u[0] - GRAV;
        return 0;
    case ORACLE_HUMAN_6D:
        o[0] = x[3] * cos(u[0]); o[1] = x[3] * sin(u[0]); o[2] = 0.0; o[3] = u[1]; o[4] = 0.0; o[5] = 0.0;
        return 0;
    case ORACLE_HUMAN_LIN_6D:
        o[0] = x[3]; o[1] = x[4]; o[2] = 0.0; o[3] = u[0]; o[4] = u[1]; o[5] = 0.0;
        return 0;
    case ORACLE_QUADCOPTER_12D: {
        const double sps = sin(x[3]), cps = cos(x[3]); /* psi   */
        const double sth = sin(x[4]), cth = cos(x[4]); /* theta */
        const double sph = sin(x[5]), cph = cos(x[5]); /* phi   */
        const double tth = tan(x[4]);
        const double vx = x[6], vy = x[7], vz = x[8], wx = x[9], wy = x[10], wz = x[11];
        o[0] = vx * cps * cth + vy * (sph * sth * cps - sps * cph) + vz * (sph * sps + sth * cph * cps);
        o[1] = vx * sps * cth + vy * (sph * sps * sth + cph * cps) + vz * (-sph * cps + sps * sth * cph);
        o[2] = -vx * sth + vy * sph * cth + vz * cph * cth;
        o[3] = wy * sph / cth + wz * cph / cth;
        o[4] = wy * cph - wz * sph;
        o[5] = wx + wy * sph * tth + wz * cph * tth;
        o[6] = vy * wz - vz * wy + GRAV * sth;
        o[7] = -vx * wz + vz * wx - GRAV * sph * cth;
        o[8] = Q12_FZ * u[3] + vx * wy - vy * wx - GRAV * cph * cth;
        o[9] = Q12_TX * u[0] - Q12_CX * wy * wz;
        o[10] = Q12_TY * u[1] + Q12_CY * wx * wz;
        o[11] = Q12_TZ * u[2] - Q12_CZ * wx * wy;
        return 0;
    }
    case ORACLE_HUMAN_PAD_12D: {
        /* BASELINE config 5's zero-padded human (not a reference model: the reference cannot stack 12- and 6-state
         * agents, dynamics.py:165-170): HumanDynamics6D (cpp:308-329) in the first 6 states / 3 controls, the
         * padded states do not move, the padded control does nothing.  Pinned by tests/golden/g8_hetero_*.npz,
         * which a shim class made of reference calls produced (tests/golden/make_golden.py). */
        int i;
        o[0] = x[3] * cos(u[0]); o[1] = x[3] * sin(u[0]); o[2] = 0.0; o[3] = u[1]; o[4] = 0.0; o[5] = 0.0;
        for (i = 6; i < 12; ++i) o[i] = 0.0;
        return 0;
    }
    default:
        return -1;
    }
}

/* classical RK4 with 5 fixed sub-steps, bbdynamics.cpp:39-93 */
int oracle_model_integrate(int model, const double *x, const double *u, double dt, double *xn)
{
    int n_s, n_c, i, j;
    double k0[MAX_NS], k1[MAX_NS], k2[MAX_NS], k3[MAX_NS], xa[MAX_NS], xb[MAX_NS];
    if (oracle_model_dims(model, &n_s, &n_c)) return -1;
    const int n_sub = 5;
    const double dh = dt / n_sub;
    for (i = 0; i < n_s; ++i) xn[i] = x[i];
    for (j = 0; j < n_sub; ++j) {
        for (i = 0; i < n_s; ++i) xa[i] = xn[i];
        oracle_model_f(model, xa, u, k0);
        for (i = 0; i < n_s; ++i) xb[i] = xa[i] + (dh / 2.0) * k0[i];
        oracle_model_f(model, xb, u, k1);
        for (i = 0; i < n_s; ++i) xb[i] = xa[i] + (dh / 2.0) * k1[i];
        oracle_model_f(model, xb, u, k2);
        for (i = 0; i < n_s; ++i) xb[i] = xa[i] + dh * k2[i];
        oracle_model_f(model, xb, u, k3);
        for (i = 0; i < n_s; ++i) xn[i] += dh * (k0[i] + 2.0 * k1[i] + 2.0 * k2[i] + k3[i]) / 6.0;
    }
    return 0;
}

/* continuous Jacobians (bbdynamics.cpp:119-148,163-228,241-262,276-306,331-391,408-415,431-491,513-711)
 * followed by the forward-Euler discretisation A <- I + dt*A, B <- dt*B (cpp:95-106, quirk Q4) */
int oracle_model_linearize(int model, const double *x, const double *u, double dt, double *A, double *B)
{
    int n_s, n_c, i;
    if (oracle_model_dims(model, &n_s, &n_c)) return -1;
    memset(A, 0, sizeof(double) * n_s * n_s);
    memset(B, 0, sizeof(double) * n_s * n_c);
#define AA(r, c) A[(r) * n_s + (c)]
#define BB(r, c) B[(r) * n_c + (c)]
    switch (model) {
    case ORACLE_DOUBLE_INT_4D:
        AA(0, 2) = 1; AA(1, 3) = 1; BB(2, 0) = 1; BB(3, 1) = 1;
        break;
    case ORACLE_DOUBLE_INT_6D:
    case ORACLE_HUMAN_LIN_6D:
        AA(0, 3) = 1; AA(1, 4) = 1; AA(2, 5) = 1; BB(3, 0) = 1; BB(4, 1) = 1; BB(5, 2) = 1;
        break;
    case ORACLE_CAR_3D:
        AA(0, 2) = -u[0] * sin(x[2]); AA(1, 2) = u[0] * cos(x[2]);
        BB(0, 0) = cos(x[2]); BB(1, 0) = sin(x[2]); BB(2, 1) = 1;
        break;
    case ORACLE_UNICYCLE_4D:
        AA(0, 2) = cos(x[3]); AA(0, 3) = -x[2] * sin(x[3]);
        AA(1, 2) = sin(x[3]); AA(1, 3) = x[2] * cos(x[3]);
        BB(2, 0) = 1; BB(3, 1) = 1;
        break;
    case ORACLE_QUADCOPTER_6D: {
        const double t2 = tan(u[2]), t1 = tan(u[1]);
        AA(0, 3) = 1; AA(1, 4) = 1; AA(2, 5) = 1;
        BB(3, 2) = GRAV * (t2 * t2) + GRAV;
        BB(4, 1) = -GRAV * (t1 * t1) - GRAV;
        BB(5, 0) = 1;
        break;
    }
    case ORACLE_HUMAN_6D:
    case ORACLE_HUMAN_PAD_12D: /* same entries in the leading 6 x 6 / 6 x 3 corner; the padding's A_c, B_c are zero */
        AA(0, 3) = cos(u[0]); AA(1, 3) = sin(u[0]);
        BB(0, 0) = -x[3] * sin(u[0]); BB(1, 0) = x[3] * cos(u[0]); BB(3, 1) = 1;
        break;
    case ORACLE_QUADCOPTER_12D: {
        const double sps = sin(x[3]), cps = cos(x[3]);
        const double sth = sin(x[4]), cth = cos(x[4]);
        const double sph = sin(x[5]), cph = cos(x[5]);
        const double tth = tan(x[4]);
        const double c2 = cth * cth;          /* pow(cos,2) */
        const double sec2 = tth * tth + 1;    /* pow(tan,2)+1 */
        const double vx = x[6], vy = x[7], vz = x[8], wx = x[9], wy = x[10], wz = x[11];
        /* d(position rate)/d(angles, body velocity) */
        AA(0, 3) = -vx * sps * cth + vy * (-sph * sps * sth - cph * cps) + vz * (sph * cps - sps * sth * cph);
        AA(0, 4) = -vx * sth * cps + vy * sph * cps * cth + vz * cph * cps * cth;
        AA(0, 5) = vy * (sph * sps + sth * cph * cps) + vz * (-sph * sth * cps + sps * cph);
        AA(0, 6) = cps * cth;
        AA(0, 7) = sph * sth * cps - sps * cph;
        AA(0, 8) = sph * sps + sth * cph * cps;
        AA(1, 3) = vx * cps * cth + vy * (sph * sth * cps - sps * cph) + vz * (sph * sps + sth * cph * cps);
        AA(1, 4) = -vx * sps * sth + vy * sph * sps * cth + vz * sps * cph * cth;
        AA(1, 5) = vy * (-sph * cps + sps * sth * cph) + vz * (-sph * sps * sth - cph * cps);
        AA(1, 6) = sps * cth;
        AA(1, 7) = sph * sps * sth + cph * cps;
        AA(1, 8) = -sph * cps + sps * sth * cph;
        AA(2, 4) = -vx * cth - vy * sph * sth - vz * sth * cph;
        AA(2, 5) = vy * cph * cth - vz * sph * cth;
        AA(2, 6) = -sth;
        AA(2, 7) = sph * cth;
        AA(2, 8) = cph * cth;
        /* Euler-angle kinematics */
        AA(3, 4) = wy * sph * sth / c2 + wz * sth * cph / c2;
        AA(3, 5) = wy * cph / cth - wz * sph / cth;
        AA(3, 10) = sph / cth;
        AA(3, 11) = cph / cth;
        AA(4, 5) = -wy * sph - wz * cph;
        AA(4, 10) = cph;
        AA(4, 11) = -sph;
        AA(5, 4) = wy * sec2 * sph + wz * sec2 * cph;
        AA(5, 5) = wy * cph * tth - wz * sph * tth;
        AA(5, 9) = 1;
        AA(5, 10) = sph * tth;
        AA(5, 11) = cph * tth;
        /* body-frame accelerations */
        AA(6, 4) = GRAV * cth; AA(6, 7) = wz; AA(6, 8) = -wy; AA(6, 10) = -vz; AA(6, 11) = vy;
        AA(7, 4) = GRAV * sph * sth; AA(7, 5) = -GRAV * cph * cth;
        AA(7, 6) = -wz; AA(7, 8) = wx; AA(7, 9) = vz; AA(7, 11) = -vx;
        AA(8, 4) = GRAV * sth * cph; AA(8, 5) = GRAV * sph * cth;
        AA(8, 6) = wy; AA(8, 7) = -wx; AA(8, 9) = -vy; AA(8, 10) = vx;
        /* rigid-body rates */
        AA(9, 10) = -Q12_CX * wz; AA(9, 11) = -Q12_CX * wy;
        AA(10, 9) = Q12_CY * wz; AA(10, 11) = Q12_CY * wx;
        AA(11, 9) = -Q12_CZ * wy; AA(11, 10) = -Q12_CZ * wx;
        BB(8, 3) = Q12_FZ; BB(9, 0) = Q12_TX; BB(10, 1) = Q12_TY; BB(11, 2) = Q12_TZ;
        break;
    }
    default:
        return -1;
    }
    /* euler_method_discretization, cpp:95-106 */
    for (i = 0; i < n_s * n_s; ++i) {
        A[i] *= dt;
        if (i % (n_s + 1) == 0) A[i] += 1;
    }
    for (i = 0; i < n_s * n_c; ++i) B[i] *= dt;
    if (model == ORACLE_HUMAN_LIN_6D) { /* cpp:408-415: no motion along z */
        AA(2, 5) = 0;
        BB(5, 2) = 0;
    }
#undef AA
#undef BB
    return 0;
}

/* ------------------------------------------------------------------- costs */

/* cost.py:269-315 */
void oracle_quadraticize_distance(const double *pa, const double *pb, double radius, int n_d,
                                  double *g, double *H)
{
    double a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
    int i;
    for (i = 0; i < n_d; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
    for (i = 0; i < 3; ++i) g[i] = 0.0;
    for (i = 0; i < 9; ++i) H[i] = 0.0;
    const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    const double dist = sqrt(dx * dx + dy * dy + dz * dz);
    if (dist > radius) return; /* active iff not (distance > radius), quirk Q7 */
    const double gs = 2 * (dist - radius) / dist;
    double gg[3] = {gs * dx, gs * dy, gs * dz};
    /* cross factor recomputes the distance as |a|^2+|b|^2-2a.b (cost.py:293-303) */
    const double h2a = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    const double h2b = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    const double dalt = sqrt((h2a + h2b) - 2 * (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]));
    const double cross = 2 * radius / pow(dalt, 3);
    const double d3 = pow(dist, 3);
    double HH[9];
    const double dd[3] = {dx, dy, dz};
    for (i = 0; i < 9; ++i) HH[i] = 0.0;
    for (i = 0; i < 3; ++i) HH[i * 3 + i] = 2 * radius * (dd[i] * dd[i]) / d3 - 2 * radius / dist + 2;
    HH[0 * 3 + 1] = HH[1 * 3 + 0] = (dx * dy) * cross;
    HH[0 * 3 + 2] = HH[2 * 3 + 0] = (dx * dz) * cross;
    HH[1 * 3 + 2] = HH[2 * 3 + 1] = (dy * dz) * cross;
    for (i = 0; i < n_d; ++i) {
        g[i] = gg[i];
        for (int j = 0; j < n_d; ++j) H[i * 3 + j] = HH[i * 3 + j];
    }
}

static int homogeneous_ndims(const oracle_problem *p)
{
    for (int i = 1; i < p->k; ++i)
        if (p->n_dims[i] != p->n_dims[0]) return 0;
    return 1;
}

/* ProximityCost.__call__, cost.py:117-133 (+ util.py:48-87).  Homogeneous
 * n_dims -> PLANAR distance (compute_pairwise_distance default n_d=2, Q5). */
double oracle_prox_cost(const oracle_problem *p, const double *x)
{
    if (p->k == 1) return 0.0;
    const int homog = homogeneous_ndims(p);
    double total = 0.0;
    for (int i = 0; i < p->k; ++i)
        for (int j = i + 1; j < p->k; ++j) {
            int nd = homog ? 2 : (p->n_dims[i] < p->n_dims[j] ? p->n_dims[i] : p->n_dims[j]);
            double s = 0.0;
            for (int c = 0; c < nd; ++c) {
                double df = x[i * p->n_s + c] - x[j * p->n_s + c];
                s += df * df;
            }
            double m = fmin(0.0, sqrt(s) - p->radius);
            total += m * m;
        }
    return total;
}

/* ProximityCost.quadraticize, cost.py:135-171 */
void oracle_prox_quadraticize(const oracle_problem *p, const double *x, double *Lx, double *Lxx)
{
    const int n_x = p->k * p->n_s;
    memset(Lx, 0, sizeof(double) * n_x);
    memset(Lxx, 0, sizeof(double) * n_x * n_x);
    for (int i = 0; i < p->k; ++i)
        for (int j = i + 1; j < p->k; ++j) {
            const int nd = p->n_dims[i] < p->n_dims[j] ? p->n_dims[i] : p->n_dims[j];
            const int ix = p->n_s * i, jx = p->n_s * j;
            double g[3], H[9];
            oracle_quadraticize_distance(x + ix, x + jx, p->radius, nd, g, H);
            for (int a = 0; a < nd; ++a) {
                Lx[ix + a] += g[a];
                Lx[jx + a] += -g[a];
                for (int b = 0; b < nd; ++b) {
                    Lxx[(ix + a) * n_x + ix + b] += H[a * 3 + b];
                    Lxx[(jx + a) * n_x + jx + b] += H[a * 3 + b];
                    Lxx[(ix + a) * n_x + jx + b] += -H[a * 3 + b];
                    Lxx[(jx + a) * n_x + ix + b] += -H[a * 3 + b];
                }
            }
        }
}

/* ReferenceCost.__call__, cost.py:79-83: (e @ Q) @ e + (u @ R) @ u */
static double ref_cost(const oracle_problem *p, int a, const double *x, const double *u, int terminal)
{
    const int ns = p->n_s, nc = p->n_c;
    const double *M = terminal ? p->Qf + a * ns * ns : p->Q + a * ns * ns;
    const double *xf = p->xf + a * ns;
    double e[MAX_NS], c = 0.0;
    for (int i = 0; i < ns; ++i) e[i] = x[i] - xf[i];
    for (int j = 0; j < ns; ++j) {
        double v = 0.0;
        for (int i = 0; i < ns; ++i) v += e[i] * M[i * ns + j];
        c += v * e[j];
    }
    if (terminal) return c;
    const double *Rm = p->R + a * nc * nc;
    double cu = 0.0;
    for (int j = 0; j < nc; ++j) {
        double v = 0.0;
        for (int i = 0; i < nc; ++i) v += u[i] * Rm[i * nc + j];
        cu += v * u[j];
    }
    return c + cu;
}

/* GameCost.__call__, cost.py:197-206 */
double oracle_cost(const oracle_problem *p, const double *x, const double *u, int terminal)
{
    double ref_total = 0.0;
    for (int a = 0; a < p->k; ++a)
        ref_total += ref_cost(p, a, x + a * p->n_s, u + a * p->n_c, terminal);
    return p->w_prox * oracle_prox_cost(p, x) + p->w_ref * ref_total;
}

/* GameCost.quadraticize, cost.py:208-239 with ReferenceCost.quadraticize :85-101 */
void oracle_quadraticize(const oracle_problem *p, const double *x, const double *u, int terminal,
                         double *Lx, double *Lu, double *Lxx, double *Luu, double *Lux)
{
    const int ns = p->n_s, nc = p->n_c, k = p->k, n_x = k * ns, n_u = k * nc;
    memset(Lx, 0, sizeof(double) * n_x);
    memset(Lu, 0, sizeof(double) * n_u);
    memset(Lxx, 0, sizeof(double) * n_x * n_x);
    memset(Luu, 0, sizeof(double) * n_u * n_u);
    memset(Lux, 0, sizeof(double) * n_u * n_x);
    for (int a = 0; a < k; ++a) {
        const double *M = terminal ? p->Qf + a * ns * ns : p->Q + a * ns * ns;
        const double *Rm = p->R + a * nc * nc;
        double e[MAX_NS];
        for (int i = 0; i < ns; ++i) e[i] = x[a * ns + i] - p->xf[a * ns + i];
        for (int j = 0; j < ns; ++j) {
            double v = 0.0;
            for (int i = 0; i < ns; ++i) v += e[i] * (M[i * ns + j] + M[j * ns + i]);
            Lx[a * ns + j] = p->w_ref * v;
            for (int i = 0; i < ns; ++i)
                Lxx[(a * ns + i) * n_x + a * ns + j] = p->w_ref * (M[i * ns + j] + M[j * ns + i]);
        }
        if (!terminal) {
            for (int j = 0; j < nc; ++j) {
                double v = 0.0;
                for (int i = 0; i < nc; ++i) v += u[a * nc + i] * (Rm[i * nc + j] + Rm[j * nc + i]);
                Lu[a * nc + j] = p->w_ref * v;
                for (int i = 0; i < nc; ++i)
                    Luu[(a * nc + i) * n_u + a * nc + j] = p->w_ref * (Rm[i * nc + j] + Rm[j * nc + i]);
            }
        }
    }
    if (k > 1) {
        double *gx = (double *)malloc(sizeof(double) * n_x);
        double *gxx = (double *)malloc(sizeof(double) * n_x * n_x);
        oracle_prox_quadraticize(p, x, gx, gxx);
        for (int i = 0; i < n_x; ++i) Lx[i] += p->w_prox * gx[i];
        for (int i = 0; i < n_x * n_x; ++i) Lxx[i] += p->w_prox * gxx[i];
        free(gx);
        free(gxx);
    }
}

/* ---------------------------------------------------------- joint dynamics */

/* MultiDynamicalModel.__call__, dynamics.py:159-171 */
void oracle_step(const oracle_problem *p, const double *x, const double *u, double *xn)
{
    for (int a = 0; a < p->k; ++a)
        oracle_model_integrate(p->model[a], x + a * p->n_s, u + a * p->n_c, p->dt, xn + a * p->n_s);
}

/* MultiDynamicalModel.linearize, dynamics.py:173-186 (uniform_block_diag, util.py:229-236) */
void oracle_linearize(const oracle_problem *p, const double *x, const double *u, double *A, double *B)
{
    const int ns = p->n_s, nc = p->n_c, n_x = p->k * ns, n_u = p->k * nc;
    double Aa[MAX_NS * MAX_NS], Ba[MAX_NS * MAX_NS];
    memset(A, 0, sizeof(double) * n_x * n_x);
    memset(B, 0, sizeof(double) * n_x * n_u);
    for (int a = 0; a < p->k; ++a) {
        oracle_model_linearize(p->model[a], x + a * ns, u + a * nc, p->dt, Aa, Ba);
        for (int i = 0; i < ns; ++i) {
            for (int j = 0; j < ns; ++j) A[(a * ns + i) * n_x + a * ns + j] = Aa[i * ns + j];
            for (int j = 0; j < nc; ++j) B[(a * ns + i) * n_u + a * nc + j] = Ba[i * nc + j];
        }
    }
}

/* ------------------------------------------------------------------ passes */

/* ilqrSolver._rollout, control.py:80-93 */
double oracle_rollout(const oracle_problem *p, const double *x0, const double *U, double *X)
{
    const int n_x = p->k * p->n_s, n_u = p->k * p->n_c, T = p->T;
    double J = 0.0;
    memcpy(X, x0, sizeof(double) * n_x);
    for (int t = 0; t < T; ++t) {
        oracle_step(p, X + t * n_x, U + t * n_u, X + (t + 1) * n_x);
        J += oracle_cost(p, X + t * n_x, U + t * n_u, 0);
    }
    double *zu = (double *)calloc(n_u, sizeof(double));
    J += oracle_cost(p, X + T * n_x, zu, 1);
    free(zu);
    return J;
}

/* C[m x n] = A^T[m x k] B[k x n], A stored k x m */
/* (loop order i, l, j: every C[i][j] still accumulates its k products in the order l = 0, 1, ... from 0.0 -- the same
 * roundings as a dot-product loop -- but the inner loop runs along rows of B and C, so the compiler can vectorise it:
 * the 240-state passes of config 5 cost 7 s each otherwise) */
#define VEC __attribute__((optimize("O3", "tree-vectorize")))
VEC static void mm_tn(int m, int n, int k, const double *restrict A, const double *restrict B, double *restrict C)
{
    for (int i = 0; i < m; ++i) {
        double *restrict c = C + (size_t)i * n;
        for (int j = 0; j < n; ++j) c[j] = 0.0;
        for (int l = 0; l < k; ++l) {
            const double a = A[l * m + i];
            const double *restrict b = B + (size_t)l * n;
            for (int j = 0; j < n; ++j) c[j] += a * b[j];
        }
    }
}
/* C[m x n] = A[m x k] B[k x n] */
VEC static void mm_nn(int m, int n, int k, const double *restrict A, const double *restrict B, double *restrict C)
{
    for (int i = 0; i < m; ++i) {
        double *restrict c = C + (size_t)i * n;
        for (int j = 0; j < n; ++j) c[j] = 0.0;
        for (int l = 0; l < k; ++l) {
            const double a = A[i * k + l];
            const double *restrict b = B + (size_t)l * n;
            for (int j = 0; j < n; ++j) c[j] += a * b[j];
        }
    }
}

/* Solve M X = R for nrhs right-hand sides by LU with partial (row) pivoting, the
 * algorithm of LAPACK dgesv behind np.linalg.solve (control.py:141-142).
 * M [m x m] and R [m x nrhs] are overwritten; returns -1 on an exactly zero pivot. */
VEC static int lu_solve(int m, int nrhs, double *restrict M, double *restrict R)
{
    for (int c = 0; c < m; ++c) {
        int piv = c;
        double best = fabs(M[c * m + c]);
        for (int r = c + 1; r < m; ++r)
            if (fabs(M[r * m + c]) > best) { best = fabs(M[r * m + c]); piv = r; }
        if (best == 0.0) return -1;
        if (piv != c) {
            for (int j = 0; j < m; ++j) { double t = M[c * m + j]; M[c * m + j] = M[piv * m + j]; M[piv * m + j] = t; }
            for (int j = 0; j < nrhs; ++j) { double t = R[c * nrhs + j]; R[c * nrhs + j] = R[piv * nrhs + j]; R[piv * nrhs + j] = t; }
        }
        const double inv = 1.0 / M[c * m + c];
        for (int r = c + 1; r < m; ++r) {
            const double l = M[r * m + c] * inv;
            M[r * m + c] = l;
            for (int j = c + 1; j < m; ++j) M[r * m + j] -= l * M[c * m + j];
            for (int j = 0; j < nrhs; ++j) R[r * nrhs + j] -= l * R[c * nrhs + j];
        }
    }
    for (int r = m - 1; r >= 0; --r)
        for (int j = 0; j < nrhs; ++j) {
            double s = R[r * nrhs + j];
            for (int c = r + 1; c < m; ++c) s -= M[r * m + c] * R[c * nrhs + j];
            R[r * nrhs + j] = s / M[r * m + r];
        }
    return 0;
}

typedef struct {
    double *AtP, *BtP, *Preg, *Qxx, *Quu, *Qux, *Qx, *Qu, *LU, *RHS, *KtQuu, *tmpnn, *tmpnn2, *p, *P, *Pn;
} bwd_ws;

static void ws_alloc(bwd_ws *w, int n, int m)
{
    w->AtP = (double *)malloc(sizeof(double) * n * n);
    w->BtP = (double *)malloc(sizeof(double) * m * n);
    w->Preg = (double *)malloc(sizeof(double) * n * n);
    w->Qxx = (double *)malloc(sizeof(double) * n * n);
    w->Quu = (double *)malloc(sizeof(double) * m * m);
    w->Qux = (double *)malloc(sizeof(double) * m * n);
    w->Qx = (double *)malloc(sizeof(double) * n);
    w->Qu = (double *)malloc(sizeof(double) * m);
    w->LU = (double *)malloc(sizeof(double) * m * m);
    w->RHS = (double *)malloc(sizeof(double) * m * (n + 1));
    w->KtQuu = (double *)malloc(sizeof(double) * n * m);
    w->tmpnn = (double *)malloc(sizeof(double) * n * n);
    w->tmpnn2 = (double *)malloc(sizeof(double) * n * n);
    w->p = (double *)malloc(sizeof(double) * n);
    w->P = (double *)malloc(sizeof(double) * n * n);
    w->Pn = (double *)malloc(sizeof(double) * n * n);
}
static void ws_free(bwd_ws *w)
{
    free(w->AtP); free(w->BtP); free(w->Preg); free(w->Qxx); free(w->Quu); free(w->Qux); free(w->Qx);
    free(w->Qu); free(w->LU); free(w->RHS); free(w->KtQuu); free(w->tmpnn); free(w->tmpnn2);
    free(w->p); free(w->P); free(w->Pn);
}

/* One step of the Riccati recursion, control.py:132-146.  p,P in ws are updated. */
static int riccati_step(int n, int m, double mu, const double *A, const double *B, const double *Lx,
                        const double *Lu, const double *Lxx, const double *Luu, const double *Lux,
                        bwd_ws *w, double *K, double *d)
{
    int i, j;
    double *p = w->p, *P = w->P;
    /* Q_x = L_x + A^T p ; Q_u = L_u + B^T p */
    for (i = 0; i < n; ++i) {
        double s = 0.0;
        for (j = 0; j < n; ++j) s += A[j * n + i] * p[j];
        w->Qx[i] = Lx[i] + s;
    }
    for (i = 0; i < m; ++i) {
        double s = 0.0;
        for (j = 0; j < n; ++j) s += B[j * m + i] * p[j];
        w->Qu[i] = Lu[i] + s;
    }
    /* Q_xx = L_xx + (A^T P) A */
    mm_tn(n, n, n, A, P, w->AtP);
    mm_nn(n, n, n, w->AtP, A, w->tmpnn);
    for (i = 0; i < n * n; ++i) w->Qxx[i] = Lxx[i] + w->tmpnn[i];
    /* reg = mu*I added to P inside Q_uu and Q_ux only (quirk Q6) */
    memcpy(w->Preg, P, sizeof(double) * n * n);
    for (i = 0; i < n; ++i) w->Preg[i * n + i] += mu;
    mm_tn(m, n, n, B, w->Preg, w->BtP);
    mm_nn(m, m, n, w->BtP, B, w->tmpnn);
    for (i = 0; i < m * m; ++i) w->Quu[i] = Luu[i] + w->tmpnn[i];
    mm_nn(m, n, n, w->BtP, A, w->tmpnn);
    for (i = 0; i < m * n; ++i) w->Qux[i] = Lux[i] + w->tmpnn[i];
    /* K = -solve(Q_uu, Q_ux) ; d = -solve(Q_uu, Q_u) */
    memcpy(w->LU, w->Quu, sizeof(double) * m * m);
    for (i = 0; i < m; ++i) {
        for (j = 0; j < n; ++j) w->RHS[i * (n + 1) + j] = w->Qux[i * n + j];
        w->RHS[i * (n + 1) + n] = w->Qu[i];
    }
    if (lu_solve(m, n + 1, w->LU, w->RHS)) return -1;
    for (i = 0; i < m; ++i) {
        for (j = 0; j < n; ++j) K[i * n + j] = -w->RHS[i * (n + 1) + j];
        d[i] = -w->RHS[i * (n + 1) + n];
    }
    /* p = Q_x + (K^T Q_uu) d + K^T Q_u + Q_ux^T d */
    mm_tn(n, m, m, K, w->Quu, w->KtQuu);
    for (i = 0; i < n; ++i) {
        double s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (j = 0; j < m; ++j) s1 += w->KtQuu[i * m + j] * d[j];
        for (j = 0; j < m; ++j) s2 += K[j * n + i] * w->Qu[j];
        for (j = 0; j < m; ++j) s3 += w->Qux[j * n + i] * d[j];
        w->tmpnn2[i] = ((w->Qx[i] + s1) + s2) + s3;
    }
    memcpy(p, w->tmpnn2, sizeof(double) * n);
    /* P = Q_xx + (K^T Q_uu) K + K^T Q_ux + Q_ux^T K ; P = (P + P^T)/2 */
    mm_nn(n, n, m, w->KtQuu, K, w->tmpnn);
    for (i = 0; i < n * n; ++i) w->Pn[i] = w->Qxx[i] + w->tmpnn[i];
    mm_tn(n, n, m, K, w->Qux, w->tmpnn);
    for (i = 0; i < n * n; ++i) w->Pn[i] += w->tmpnn[i];
    mm_tn(n, n, m, w->Qux, K, w->tmpnn);
    for (i = 0; i < n * n; ++i) w->Pn[i] += w->tmpnn[i];
    for (i = 0; i < n; ++i)
        for (j = 0; j < n; ++j) P[i * n + j] = 0.5 * (w->Pn[i * n + j] + w->Pn[j * n + i]);
    return 0;
}

/* ilqrSolver._backward_pass, control.py:116-148 */
int oracle_backward_pass(const oracle_problem *p, const double *X, const double *U, double mu,
                         double *K, double *d)
{
    const int n = p->k * p->n_s, m = p->k * p->n_c, T = p->T;
    bwd_ws w;
    ws_alloc(&w, n, m);
    double *A = (double *)malloc(sizeof(double) * n * n), *B = (double *)malloc(sizeof(double) * n * m);
    double *Lx = (double *)malloc(sizeof(double) * n), *Lu = (double *)malloc(sizeof(double) * m);
    double *Lxx = (double *)malloc(sizeof(double) * n * n), *Luu = (double *)malloc(sizeof(double) * m * m);
    double *Lux = (double *)malloc(sizeof(double) * m * n), *zu = (double *)calloc(m, sizeof(double));
    int rc = 0;
    oracle_quadraticize(p, X + T * n, zu, 1, w.p, Lu, w.P, Luu, Lux);
    for (int t = T - 1; t >= 0 && !rc; --t) {
        oracle_quadraticize(p, X + t * n, U + t * m, 0, Lx, Lu, Lxx, Luu, Lux);
        oracle_linearize(p, X + t * n, U + t * m, A, B);
        rc = riccati_step(n, m, mu, A, B, Lx, Lu, Lxx, Luu, Lux, &w, K + (size_t)t * m * n, d + t * m);
    }
    free(A); free(B); free(Lx); free(Lu); free(Lxx); free(Luu); free(Lux); free(zu);
    ws_free(&w);
    return rc;
}

/* Same recursion on explicit tiles: A[T][n][n], B[T][n][m], Lx[T+1][n], Lu[T+1][m],
 * Lxx[T+1][n][n], Luu[T+1][m][m], Lux[T+1][m][n]; entry T is the terminal quadraticisation. */
int oracle_backward_pass_tiles(int n, int m, int T, const double *A, const double *B, const double *Lx,
                               const double *Lu, const double *Lxx, const double *Luu, const double *Lux,
                               double mu, double *K, double *d)
{
    bwd_ws w;
    int rc = 0;
    ws_alloc(&w, n, m);
    memcpy(w.p, Lx + (size_t)T * n, sizeof(double) * n);
    memcpy(w.P, Lxx + (size_t)T * n * n, sizeof(double) * n * n);
    for (int t = T - 1; t >= 0 && !rc; --t)
        rc = riccati_step(n, m, mu, A + (size_t)t * n * n, B + (size_t)t * n * m, Lx + (size_t)t * n,
                          Lu + (size_t)t * m, Lxx + (size_t)t * n * n, Luu + (size_t)t * m * m,
                          Lux + (size_t)t * m * n, &w, K + (size_t)t * m * n, d + (size_t)t * m);
    ws_free(&w);
    return rc;
}

/* ilqrSolver._forward_pass, control.py:95-114 */
double oracle_forward_pass(const oracle_problem *p, const double *X, const double *U, const double *K,
                           const double *d, double alpha, double *Xn, double *Un)
{
    const int n = p->k * p->n_s, m = p->k * p->n_c, T = p->T;
    double J = 0.0;
    double *dx = (double *)malloc(sizeof(double) * n);
    memcpy(Xn, X, sizeof(double) * n);
    for (int t = 0; t < T; ++t) {
        for (int i = 0; i < n; ++i) dx[i] = Xn[t * n + i] - X[t * n + i];
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += K[((size_t)t * m + i) * n + j] * dx[j];
            const double du = s + alpha * d[t * m + i];
            Un[t * m + i] = U[t * m + i] + du;
        }
        oracle_step(p, Xn + t * n, Un + t * m, Xn + (t + 1) * n);
        J += oracle_cost(p, Xn + t * n, Un + t * m, 0);
    }
    double *zu = (double *)calloc(m, sizeof(double));
    J += oracle_cost(p, Xn + T * n, zu, 1);
    free(zu);
    free(dx);
    return J;
}

/* 1.1 ** (-arange(10, dtype=float32) ** 2), control.py:162: float32 values (quirk Q1) */
void oracle_alphas(double *a)
{
    static const unsigned int bits[10] = {0x3f800000u, 0x3f68ba2eu, 0x3f2ed9f7u, 0x3ed92350u, 0x3e5eda27u,
                                          0x3dbd05a8u, 0x3d04808du, 0x3c19864au, 0x3b13029cu, 0x39e8ae70u};
    for (int i = 0; i < 10; ++i) {
        float f;
        memcpy(&f, &bits[i], 4);
        a[i] = (double)f;
    }
}

/* ilqrSolver.solve, control.py:150-225 (+ regularisation schedule :227-237) */
int oracle_solve(const oracle_problem *p, const double *x0, double *U, int n_lqr_iter, double tol,
                 double *X, double *J_out, double *trace, int *n_bwd, int *n_fwd)
{
    const int n = p->k * p->n_s, m = p->k * p->n_c, T = p->T;
    double alphas[10];
    oracle_alphas(alphas);
    double mu = 1.0, delta = 2.0; /* _reset_regularization */
    double *K = (double *)malloc(sizeof(double) * (size_t)T * m * n), *d = (double *)malloc(sizeof(double) * T * m);
    double *Xn = (double *)malloc(sizeof(double) * (T + 1) * n), *Un = (double *)malloc(sizeof(double) * T * m);
    double J_star = oracle_rollout(p, x0, U, X);
    double J = J_star; /* the reference leaves J unbound when n_lqr_iter == 0 (Q2) */
    int status = ORACLE_MAX_ITER, nb = 0, nf = 0;
    for (int it = 0; it < n_lqr_iter; ++it) {
        int accept = 0, converged = 0, acc_idx = -1, nf_it = 0;
        const double mu_before = mu;
        if (oracle_backward_pass(p, X, U, mu, K, d)) { status = -1; break; }
        ++nb;
        for (int a = 0; a < 10; ++a) {
            J = oracle_forward_pass(p, X, U, K, d, alphas[a], Xn, Un);
            ++nf; ++nf_it;
            if (J < J_star) {
                if (fabs((J_star - J) / J_star) < tol) converged = 1;
                memcpy(X, Xn, sizeof(double) * (T + 1) * n);
                memcpy(U, Un, sizeof(double) * T * m);
                J_star = J;
                /* _decrease_regularization, control.py:232-237 */
                delta = fmin(1.0, delta) / 2.0;
                mu *= delta;
                if (mu <= 1e-6) mu = 0.0;
                accept = 1; acc_idx = a;
                break;
            }
        }
        if (trace) {
            trace[it * 5 + 0] = mu_before; trace[it * 5 + 1] = acc_idx; trace[it * 5 + 2] = J;
            trace[it * 5 + 3] = J_star; trace[it * 5 + 4] = nf_it;
        }
        if (!accept) { status = ORACLE_LINESEARCH_FAILED; break; } /* control.py:195-198 */
        if (converged) { status = ORACLE_CONVERGED; break; }
    }
    *J_out = J; /* last EVALUATED forward-pass cost, quirk Q2 */
    if (n_bwd) *n_bwd = nb;
    if (n_fwd) *n_fwd = nf;
    free(K); free(d); free(Xn); free(Un);
    return status;
}

/* The same loop as oracle_solve, but FOLLOWING a given decision sequence instead of taking its own: "forced" is the
 * decision trace of the implementation under test ([n_forced][5] rows as written by oracle_solve / dpilqr_solve_batch; only
 * column 1, the accepted alpha index or -1, is read) and forced_status its final status.  At every iteration the oracle
 * evaluates the candidates the implementation must have evaluated (alpha_0 .. alpha_accepted, all ten after a failed
 * search) with ITS OWN numbers, notes what it would have decided itself and how far from equality each comparison that
 * went the other way sat (control.py:183 J < J*, :184 |(J* - J)/J*| < tol), then takes the forced branch.  So every
 * iteration of the implementation's solve -- also the ones after a decision on which the two differ -- has oracle numbers
 * of the SAME iterate beside it (oracle/parity.py).  TEST INFRASTRUCTURE like everything in this file.
 *   rtrace [n_forced][8] = (mu_before, the oracle's own accepted index at this iterate (-1 none among the candidates
 *   evaluated), J of the last candidate evaluated, J* after the forced step, accept margin, convergence margin, J* before,
 *   the oracle's own converged flag).  A margin is 0 where the oracle's own verdict is the forced one, otherwise the
 *   relative distance of the comparison from equality (infinity for a NaN cost the implementation accepted).
 * Returns forced_status, or -1 if a pivot was exactly zero. */
int oracle_solve_replay(const oracle_problem *p, const double *x0, double *U, int n_lqr_iter, double tol,
                        const double *forced, int n_forced, int forced_status, double *X, double *J_out,
                        double *rtrace)
{
    const int n = p->k * p->n_s, m = p->k * p->n_c, T = p->T;
    double alphas[10];
    oracle_alphas(alphas);
    double mu = 1.0, delta = 2.0;
    double *K = (double *)malloc(sizeof(double) * (size_t)T * m * n), *d = (double *)malloc(sizeof(double) * T * m);
    double *Xn = (double *)malloc(sizeof(double) * (T + 1) * n), *Un = (double *)malloc(sizeof(double) * T * m);
    double J_star = oracle_rollout(p, x0, U, X);
    double J = J_star;
    int status = forced_status;
    if (n_forced > n_lqr_iter) n_forced = n_lqr_iter;
    for (int it = 0; it < n_forced; ++it) {
        const int a_f = (int)forced[it * 5 + 1];
        const int last = a_f >= 0 ? a_f : 9;
        const double mu_before = mu, J_before = J_star;
        double acc_margin = 0.0, conv_margin = 0.0;
        int natural = -1, nat_conv = 0;
        if (oracle_backward_pass(p, X, U, mu, K, d)) { status = -1; break; }
        for (int a = 0; a <= last; ++a) {
            J = oracle_forward_pass(p, X, U, K, d, alphas[a], Xn, Un);
            const int takes = J < J_star;
            if (takes && natural < 0) natural = a;
            if (a != a_f && takes) {                       /* the implementation rejected a candidate the oracle takes */
                const double g = (J_star - J) / fabs(J_star);
                if (g > acc_margin) acc_margin = g;
            }
            if (a == a_f && !takes) {                      /* ... accepted one the oracle rejects */
                const double g = J == J ? (J - J_star) / fabs(J_star) : INFINITY;
                if (g > acc_margin || g != g) acc_margin = g;
            }
        }
        if (a_f >= 0) {
            const double rel = fabs((J_star - J) / J_star);
            nat_conv = rel < tol;
            const int forced_conv = (it == n_forced - 1) && forced_status == ORACLE_CONVERGED;
            if (nat_conv != forced_conv) conv_margin = fabs(rel - tol);
            if (nat_conv != forced_conv && conv_margin == 0.0) conv_margin = 1e-300;   /* a flip that sat exactly on tol */
            memcpy(X, Xn, sizeof(double) * (T + 1) * n);
            memcpy(U, Un, sizeof(double) * T * m);
            J_star = J;
            delta = fmin(1.0, delta) / 2.0;
            mu *= delta;
            if (mu <= 1e-6) mu = 0.0;
        }
        rtrace[it * 8 + 0] = mu_before; rtrace[it * 8 + 1] = natural; rtrace[it * 8 + 2] = J; rtrace[it * 8 + 3] = J_star;
        rtrace[it * 8 + 4] = acc_margin; rtrace[it * 8 + 5] = conv_margin; rtrace[it * 8 + 6] = J_before;
        rtrace[it * 8 + 7] = nat_conv;
    }
    *J_out = J;
    free(K); free(d); free(Xn); free(Un);
    return status;
}

/* forced [B][n_lqr_iter][5], n_forced / forced_status [B], rtrace [B][n_lqr_iter][8] */
int oracle_replay_batch(const oracle_problem *proto, int B, const double *x0, const double *xf, double *U,
                        int n_lqr_iter, double tol, const double *forced, const int *n_forced,
                        const int *forced_status, double *X, double *J, int *status, double *rtrace, int n_threads)
{
    const int n = proto->k * proto->n_s, m = proto->k * proto->n_c, T = proto->T;
    const size_t rows = (size_t)(n_lqr_iter > 0 ? n_lqr_iter : 1);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        oracle_problem q = *proto;
        q.xf = xf + (size_t)b * n;
        status[b] = oracle_solve_replay(&q, x0 + (size_t)b * n, U + (size_t)b * T * m, n_lqr_iter, tol,
                                        forced + b * rows * 5, n_forced[b], forced_status[b],
                                        X + (size_t)b * (T + 1) * n, J + b, rtrace + b * rows * 8);
    }
    return 0;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int oracle_solve_batch(const oracle_problem *proto, int B, const double *x0, const double *xf, double *U,
                       int n_lqr_iter, double tol, double *X, double *J, int *status, int *n_bwd,
                       int *n_fwd, int n_threads)
{
    return oracle_solve_batch_trace(proto, B, x0, xf, U, n_lqr_iter, tol, X, J, status, n_bwd, n_fwd, n_threads, NULL);
}

/* the same with the per-iteration decision trace of every item: trace[B][n_lqr_iter][5] (may be NULL) */
int oracle_solve_batch_trace(const oracle_problem *proto, int B, const double *x0, const double *xf, double *U,
                             int n_lqr_iter, double tol, double *X, double *J, int *status, int *n_bwd,
                             int *n_fwd, int n_threads, double *trace)
{
    const int n = proto->k * proto->n_s, m = proto->k * proto->n_c, T = proto->T;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        oracle_problem q = *proto;
        q.xf = xf + (size_t)b * n;
        status[b] = oracle_solve(&q, x0 + (size_t)b * n, U + (size_t)b * T * m, n_lqr_iter, tol,
                                 X + (size_t)b * (T + 1) * n, J + b,
                                 trace ? trace + (size_t)b * (n_lqr_iter > 0 ? n_lqr_iter : 1) * 5 : NULL, n_bwd + b, n_fwd + b);
    }
    return 0;
}
