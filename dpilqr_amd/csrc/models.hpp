// models.hpp -- device-side dynamics for the eight reference models.
//
// Replaces the reference's C++/Cython path: bbdynamics.cpp (f_*, linearize_*, rk4,
// euler_method_discretization) reached through bbdynamicswrap.pyx:61-164.  Written from the
// model equations, not translated: each model is a struct with
//   NS, NC          per-agent state / control dimension       (dynamics.py:205-250)
//   f(x,u,o)        continuous dynamics                        (bbdynamics.cpp f_*)
//   jac(x,u,A,B)    CONTINUOUS Jacobians, zero-initialised A[NS*NS], B[NS*NC] filled sparsely
// The discretisation the solver uses is applied by the callers:
//   rollouts : classical RK4, 5 fixed sub-steps (cpp:39-93)            -> integrate<M>()
//   gains    : forward Euler  A = I + dt*A_c, B = dt*B_c (cpp:95-106)  -> linearize<M>()
// (the two deliberately differ: reference quirk Q4).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "trig_inline.hpp"

namespace dpilqr {

constexpr double kGrav = 9.80665;  // bbdynamics.cpp:11
constexpr int kMaxNs = 12;
constexpr int kMaxNc = 4;

// Model enum, identical values to bbdynamicswrap.pyx:8-16
enum Model : int {
    kDoubleInt4D = 0, kDoubleInt6D = 1, kCar3D = 2, kUnicycle4D = 3,
    kQuadcopter6D = 4, kHuman6D = 5, kHumanLin6D = 6, kQuadcopter12D = 7,
    kHumanPad12D = 8,   // this library's own: HumanDynamics6D zero-padded to 12 states / 4 controls (dpilqr_hip.h)
    kNumModels = 9
};

__host__ __device__ inline int model_ns(int m) {
    constexpr int t[kNumModels] = {4, 6, 3, 4, 6, 6, 6, 12, 12};
    return (m >= 0 && m < kNumModels) ? t[m] : -1;
}
__host__ __device__ inline int model_nc(int m) {
    constexpr int t[kNumModels] = {2, 3, 2, 2, 3, 3, 3, 4, 4};
    return (m >= 0 && m < kNumModels) ? t[m] : -1;
}

__device__ __forceinline__ void sincos_r(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ void sincos_r(float x, float* s, float* c) { sincosf(x, s, c); }

template <int M> struct ModelDef;

template <> struct ModelDef<kDoubleInt4D> {  // x=[px,py,vx,vy] u=[ax,ay]
    static constexpr int NS = 4, NC = 2;
    template <typename R> __device__ static void f(const R* x, const R* u, R* o) {
        o[0] = x[2]; o[1] = x[3]; o[2] = u[0]; o[3] = u[1];
    }
    template <typename R> __device__ static void jac(const R*, const R*, R* A, R* B) {
        A[0 * 4 + 2] = 1.0; A[1 * 4 + 3] = 1.0; B[2 * 2 + 0] = 1.0; B[3 * 2 + 1] = 1.0;
    }
};

template <> struct ModelDef<kDoubleInt6D> {  // x=[p(3),v(3)] u=[a(3)]
    static constexpr int NS = 6, NC = 3;
    template <typename R> __device__ static void f(const R* x, const R* u, R* o) {
        o[0] = x[3]; o[1] = x[4]; o[2] = x[5]; o[3] = u[0]; o[4] = u[1]; o[5] = u[2];
    }
    template <typename R> __device__ static void jac(const R*, const R*, R* A, R* B) {
        A[0 * 6 + 3] = 1.0; A[1 * 6 + 4] = 1.0; A[2 * 6 + 5] = 1.0;
        B[3 * 3 + 0] = 1.0; B[4 * 3 + 1] = 1.0; B[5 * 3 + 2] = 1.0;
    }
};

template <> struct ModelDef<kCar3D> {  // x=[px,py,theta] u=[v,omega]
    static constexpr int NS = 3, NC = 2;
    static constexpr int kHeading = 2;   // the heading angle: its rate is a control, constant over a step (see integrate)
    template <typename R> __device__ static void f_sc(const R*, const R* u, R sn, R cs, R* o) {   // f with sin / cos of the heading given
        o[0] = u[0] * cs; o[1] = u[0] * sn; o[2] = u[1];
    }
    template <typename R> __device__ static void f(const R* x, const R* u, R* o) {
        R sn, cs;
        sincos_r(x[2], &sn, &cs);
        f_sc(x, u, sn, cs, o);
    }
    template <typename R> __device__ static void jac(const R* x, const R* u, R* A, R* B) {
#ifdef DPILQR_JAC_SINCOS
        R s, c;
        sincos_r(x[2], &s, &c);
#else
        const R s = sin(x[2]), c = cos(x[2]);
#endif
        A[0 * 3 + 2] = -u[0] * s; A[1 * 3 + 2] = u[0] * c;
        B[0 * 2 + 0] = c; B[1 * 2 + 0] = s; B[2 * 2 + 1] = 1.0;
    }
};

template <> struct ModelDef<kUnicycle4D> {  // x=[px,py,v,theta] u=[a,omega]
    static constexpr int NS = 4, NC = 2;
    static constexpr int kHeading = 3;   // the heading angle: its rate is a control, constant over a step (see integrate)
    template <typename R> __device__ static void f_sc(const R* x, const R* u, R sn, R cs, R* o) {   // f with sin / cos of the heading given
        o[0] = x[2] * cs; o[1] = x[2] * sn; o[2] = u[0]; o[3] = u[1];
    }
    template <typename R> __device__ static void f(const R* x, const R* u, R* o) {
        R sn, cs;
        sincos_r(x[3], &sn, &cs);   // one argument reduction for both (same values as sin(), cos())
        f_sc(x, u, sn, cs, o);
    }
    template <typename R> __device__ static void jac(const R* x, const R*, R* A, R* B) {
#ifdef DPILQR_JAC_SINCOS
        R s, c;
        sincos_r(x[3], &s, &c);
#else
        const R s = sin(x[3]), c = cos(x[3]);
#endif
        A[0 * 4 + 2] = c; A[0 * 4 + 3] = -x[2] * s;
        A[1 * 4 + 2] = s; A[1 * 4 + 3] = x[2] * c;
        B[2 * 2 + 0] = 1.0; B[3 * 2 + 1] = 1.0;
    }
};

template <> struct ModelDef<kQuadcopter6D> {  // x=[p(3),v(3)] u=[tau,phi,theta]
    static constexpr int NS = 6, NC = 3;
    // the transcendental arguments are CONTROLS, held over a step: pre() evaluates them once, f_tr() is f given them (integrate)
    template <typename R> __device__ static void pre(const R* u, R* tr) { tr[0] = tan(u[2]); tr[1] = tan(u[1]); }
    template <typename R> __device__ static void f_tr(const R* x, const R* u, const R* tr, R* o) {
        o[0] = x[3]; o[1] = x[4]; o[2] = x[5];
        o[3] = R(kGrav) * tr[0]; o[4] = -R(kGrav) * tr[1]; o[5] = u[0] - R(kGrav);
    }
    template <typename R> __device__ static void f(const R* x, const R* u, R* o) {
        R tr[2];
        pre(u, tr);
        f_tr(x, u, tr, o);
    }
    template <typename R> __device__ static void jac(const R*, const R* u, R* A, R* B) {
        const R t2 = tan(u[2]), t1 = tan(u[1]);
        A[0 * 6 + 3] = 1.0; A[1 * 6 + 4] = 1.0; A[2 * 6 + 5] = 1.0;
        B[3 * 3 + 2] = R(kGrav) * (t2 * t2) + R(kGrav);
        B[4 * 3 + 1] = -R(kGrav) * (t1 * t1) - R(kGrav);
        B[5 * 3 + 0] = 1.0;
    }
};

template <> struct ModelDef<kHuman6D> {  // x=[px,py,pz,v,0,0] u=[heading,accel,-]
    static constexpr int NS = 6, NC = 3;
    template <typename R> __device__ static void pre(const R* u, R* tr) { sincos_r(u[0], &tr[1], &tr[0]); }
    template <typename R> __device__ static void f_tr(const R* x, const R* u, const R* tr, R* o) {
        o[0] = x[3] * tr[0]; o[1] = x[3] * tr[1]; o[2] = 0.0; o[3] = u[1]; o[4] = 0.0; o[5] = 0.0;
    }
    template <typename R> __device__ static void f(const R* x, const R* u, R* o) {
        R tr[2];
        pre(u, tr);
        f_tr(x, u, tr, o);
    }
    template <typename R> __device__ static void jac(const R* x, const R* u, R* A, R* B) {
#ifdef DPILQR_JAC_SINCOS
        R s, c;
        sincos_r(u[0], &s, &c);
#else
        const R s = sin(u[0]), c = cos(u[0]);
#endif
        A[0 * 6 + 3] = c; A[1 * 6 + 3] = s;
        B[0 * 3 + 0] = -x[3] * s; B[1 * 3 + 0] = x[3] * c; B[3 * 3 + 1] = 1.0;
    }
};

template <> struct ModelDef<kHumanLin6D> {  // planar double integrator at constant height
    static constexpr int NS = 6, NC = 3;
    template <typename R> __device__ static void f(const R* x, const R* u, R* o) {
        o[0] = x[3]; o[1] = x[4]; o[2] = 0.0; o[3] = u[0]; o[4] = u[1]; o[5] = 0.0;
    }
    // The reference discretises the DoubleInt6D Jacobians and THEN zeroes A[2][5], B[5][2]
    // (cpp:408-415); dropping the two continuous entries before discretising gives the same matrices.
    template <typename R> __device__ static void jac(const R*, const R*, R* A, R* B) {
        A[0 * 6 + 3] = 1.0; A[1 * 6 + 4] = 1.0;
        B[3 * 3 + 0] = 1.0; B[4 * 3 + 1] = 1.0;
    }
};

template <> struct ModelDef<kQuadcopter12D> {
    // x=[p(3), psi,theta,phi, v(3), w(3)]  u=[tau_x,tau_y,tau_z,f_z]; inertia constants are the
    // exact rationals of bbdynamics.cpp:507-510
    static constexpr int NS = 12, NC = 4;
    static constexpr double kFz = 2000.0 / 63.0;
    static constexpr double kTx = 625000000000000000.0 / 10982593196059.0;
    static constexpr double kTy = 5000000000000000000.0 / 92848985528431.0;
    static constexpr double kTz = 10000000000000000000.0 / 271597947137541.0;
    static constexpr double kCx = 85899976080679.0 / 175721491136944.0;
    static constexpr double kCy = 95876456000597.0 / 185697971056862.0;
    static constexpr double kCz = 9976479919918.0 / 271597947137541.0;
    template <typename R> __device__ static void f(const R* x, const R* u, R* o) {
        // (sincos: one argument reduction per angle for both values -- the same values as sin(), cos())
        // fp64, round 6: the three sincos and the tan as ONE basic block (trig_inline.hpp: the library's own small-argument
        // algorithm, bit for bit, its branch taken once for all four) -- their dependent chains interleave and tan(theta) shares
        // theta's reduction; an angle of 2^30 or more, or not finite, in any lane sends the wavefront to the library calls
        R sps, cps, sth, cth, sph, cph, tth;
        bool done = false;
        if constexpr (sizeof(R) == 8) {
            if (trig_small_all(x[3], x[4], x[5])) {
                const TrigRed r3 = trig_reduce(x[3]), r4 = trig_reduce(x[4]), r5 = trig_reduce(x[5]);
                double s_[3], c_[3];
                trig_sincos(x[3], r3, &s_[0], &c_[0]); trig_sincos(x[4], r4, &s_[1], &c_[1]); trig_sincos(x[5], r5, &s_[2], &c_[2]);
                sps = s_[0]; cps = c_[0]; sth = s_[1]; cth = c_[1]; sph = s_[2]; cph = c_[2];
                tth = trig_tan(x[4], r4);
                done = true;
            }
        }
        if (!done) {
            sincos_r(x[3], &sps, &cps); sincos_r(x[4], &sth, &cth); sincos_r(x[5], &sph, &cph);
            tth = tan(x[4]);
        }
        const R vx = x[6], vy = x[7], vz = x[8], wx = x[9], wy = x[10], wz = x[11];
        o[0] = vx * cps * cth + vy * (sph * sth * cps - sps * cph) + vz * (sph * sps + sth * cph * cps);
        o[1] = vx * sps * cth + vy * (sph * sps * sth + cph * cps) + vz * (-sph * cps + sps * sth * cph);
        o[2] = -vx * sth + vy * sph * cth + vz * cph * cth;
        o[3] = wy * sph / cth + wz * cph / cth;
        o[4] = wy * cph - wz * sph;
        o[5] = wx + wy * sph * tth + wz * cph * tth;
        o[6] = vy * wz - vz * wy + R(kGrav) * sth;
        o[7] = -vx * wz + vz * wx - R(kGrav) * sph * cth;
        o[8] = R(kFz) * u[3] + vx * wy - vy * wx - R(kGrav) * cph * cth;
        o[9] = R(kTx) * u[0] - R(kCx) * wy * wz;
        o[10] = R(kTy) * u[1] + R(kCy) * wx * wz;
        o[11] = R(kTz) * u[2] - R(kCz) * wx * wy;
    }
    template <typename R> __device__ static void jac(const R* x, const R*, R* A, R* B) {
        // (sin and cos by separate calls here: the Jacobians are evaluated once per step, their cost is nothing, and the values are
        // those of sincos.  Round 4 reverted sincos pairs because they raised the large-cluster sweep's spills from 64 to 82
        // registers, a build that died with an HSA aperture violation; since round 5 that sweep does not spill in either form
        // -- riccati_big.hpp -- and -DDPILQR_JAC_SINCOS only remains as the soak test's second build, scripts/cfg5_soak.sh.)
#ifdef DPILQR_JAC_SINCOS   // diagnostic builds only (see above)
        R sps, cps, sth, cth, sph, cph;
        sincos_r(x[3], &sps, &cps); sincos_r(x[4], &sth, &cth); sincos_r(x[5], &sph, &cph);
        const R tth = tan(x[4]);
#else
        const R sps = sin(x[3]), cps = cos(x[3]), sth = sin(x[4]), cth = cos(x[4]);
        const R sph = sin(x[5]), cph = cos(x[5]), tth = tan(x[4]);
#endif
        const R c2 = cth * cth, sec2 = tth * tth + 1;
        const R vx = x[6], vy = x[7], vz = x[8], wx = x[9], wy = x[10], wz = x[11];
#define A_(r, c) A[(r) * 12 + (c)]
        A_(0, 3) = -vx * sps * cth + vy * (-sph * sps * sth - cph * cps) + vz * (sph * cps - sps * sth * cph);
        A_(0, 4) = -vx * sth * cps + vy * sph * cps * cth + vz * cph * cps * cth;
        A_(0, 5) = vy * (sph * sps + sth * cph * cps) + vz * (-sph * sth * cps + sps * cph);
        A_(0, 6) = cps * cth;
        A_(0, 7) = sph * sth * cps - sps * cph;
        A_(0, 8) = sph * sps + sth * cph * cps;
        A_(1, 3) = vx * cps * cth + vy * (sph * sth * cps - sps * cph) + vz * (sph * sps + sth * cph * cps);
        A_(1, 4) = -vx * sps * sth + vy * sph * sps * cth + vz * sps * cph * cth;
        A_(1, 5) = vy * (-sph * cps + sps * sth * cph) + vz * (-sph * sps * sth - cph * cps);
        A_(1, 6) = sps * cth;
        A_(1, 7) = sph * sps * sth + cph * cps;
        A_(1, 8) = -sph * cps + sps * sth * cph;
        A_(2, 4) = -vx * cth - vy * sph * sth - vz * sth * cph;
        A_(2, 5) = vy * cph * cth - vz * sph * cth;
        A_(2, 6) = -sth;
        A_(2, 7) = sph * cth;
        A_(2, 8) = cph * cth;
        A_(3, 4) = wy * sph * sth / c2 + wz * sth * cph / c2;
        A_(3, 5) = wy * cph / cth - wz * sph / cth;
        A_(3, 10) = sph / cth;
        A_(3, 11) = cph / cth;
        A_(4, 5) = -wy * sph - wz * cph;
        A_(4, 10) = cph;
        A_(4, 11) = -sph;
        A_(5, 4) = wy * sec2 * sph + wz * sec2 * cph;
        A_(5, 5) = wy * cph * tth - wz * sph * tth;
        A_(5, 9) = 1.0;
        A_(5, 10) = sph * tth;
        A_(5, 11) = cph * tth;
        A_(6, 4) = R(kGrav) * cth; A_(6, 7) = wz; A_(6, 8) = -wy; A_(6, 10) = -vz; A_(6, 11) = vy;
        A_(7, 4) = R(kGrav) * sph * sth; A_(7, 5) = -R(kGrav) * cph * cth;
        A_(7, 6) = -wz; A_(7, 8) = wx; A_(7, 9) = vz; A_(7, 11) = -vx;
        A_(8, 4) = R(kGrav) * sth * cph; A_(8, 5) = R(kGrav) * sph * cth;
        A_(8, 6) = wy; A_(8, 7) = -wx; A_(8, 9) = -vy; A_(8, 10) = vx;
        A_(9, 10) = -R(kCx) * wz; A_(9, 11) = -R(kCx) * wy;
        A_(10, 9) = R(kCy) * wz; A_(10, 11) = R(kCy) * wx;
        A_(11, 9) = -R(kCz) * wy; A_(11, 10) = -R(kCz) * wx;
#undef A_
        B[8 * 4 + 3] = R(kFz); B[9 * 4 + 0] = R(kTx); B[10 * 4 + 1] = R(kTy); B[11 * 4 + 2] = R(kTz);
    }
};

template <> struct ModelDef<kHumanPad12D> {  // x=[px,py,pz,v,0,0 | 6 padded states] u=[heading,accel,-,-]
    // HumanDynamics6D in a 12-state / 4-control slot (BASELINE config 5's "zero-padded state").  The padded states
    // have x_dot = 0, so RK4 leaves them bit-for-bit where they are and the Euler Jacobians give A = 1 on their
    // diagonal and B = 0 -- what a block-diagonal embedding of the six-state model means.
    static constexpr int NS = 12, NC = 4;
    template <typename R> __device__ static void pre(const R* u, R* tr) { sincos_r(u[0], &tr[1], &tr[0]); }
    template <typename R> __device__ static void f_tr(const R* x, const R* u, const R* tr, R* o) {
        o[0] = x[3] * tr[0]; o[1] = x[3] * tr[1]; o[2] = 0.0; o[3] = u[1];
#pragma unroll
        for (int i = 4; i < 12; ++i) o[i] = 0.0;
    }
    template <typename R> __device__ static void f(const R* x, const R* u, R* o) {
        R tr[2];
        pre(u, tr);
        f_tr(x, u, tr, o);
    }
    template <typename R> __device__ static void jac(const R* x, const R* u, R* A, R* B) {
#ifdef DPILQR_JAC_SINCOS
        R s, c;
        sincos_r(u[0], &s, &c);
#else
        const R s = sin(u[0]), c = cos(u[0]);
#endif
        A[0 * 12 + 3] = c; A[1 * 12 + 3] = s;
        B[0 * 4 + 0] = -x[3] * s; B[1 * 4 + 0] = x[3] * c; B[3 * 4 + 1] = 1.0;
    }
};

// x / 6.0, correctly rounded, without the ~12-instruction dependent chain of an fp64 division: with c = RN(1/6),
// q = RN(x c), r = x - 6 q (exact in an fma), q' = RN(q + r c) is RN(x / 6) whenever the quotient is a normal number
// (Markstein's correction step; checked against the division on 6e8 random operands).  Subnormal quotients, where
// the residual is no longer exact, take the division.  The RK4 update divides by 6 four to twelve times per
// sub-step and sits on the forward pass's critical path.
__device__ __forceinline__ double div6_fast(double x) {   // exact unless the quotient is subnormal
    const double c = 0x1.5555555555555p-3;
    const double q = x * c;
    const double r = fma(-6.0, q, x);
    return fma(r, c, q);
}
// v[0..N) <- v / 6.0, correctly rounded.  One wave-uniform test covers all N operands: the binary exponent
// (v_frexp_exp: 0 for +-0) says whether any non-zero operand is small enough for a subnormal quotient, and only then is
// the fp64 division evaluated (behind a real branch: a per-lane `if` would be if-converted and the division, or an
// un-fenced one hoisted, evaluated every time).
template <int N>
__device__ __forceinline__ void div6_vec(double* v) {
    int emin = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) emin = min(emin, __builtin_amdgcn_frexp_exp(v[i]));
    double q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = div6_fast(v[i]);
    if (__builtin_amdgcn_ballot_w64(emin <= -1000) != 0ull) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double xs = v[i];
            asm volatile("" : "+v"(xs));
            const double qs = xs / 6.0;
            if (__builtin_amdgcn_frexp_exp(v[i]) <= -1000) q[i] = qs;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = q[i];
}
__device__ __forceinline__ double div6(double x) {
    double v[1] = {x};
    div6_vec<1>(v);
    return v[0];
}
// the fp32 arm (BASELINE config 5's tolerance study) divides: a single-precision division is a short sequence
template <int N>
__device__ __forceinline__ void div6_vec(float* v) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] / 6.0f;
}

// The step's divisions by 6 under ONE sub-normal guard (integrate below): MODE 1 the reciprocal form while the smallest binary
// exponent among the operands is tracked, MODE 2 the division (for the redo of a step in which some lane met an operand small
// enough for a sub-normal quotient), MODE 0 a guard per call (div6_vec: the fp32 arm, A/B builds).  Every quotient is RN(v / 6)
// whichever way: bit-identical results.
template <int MODE, int N, typename R>
__device__ __forceinline__ void div6_sel(R* v, int& emin) {
    if constexpr (MODE == 0 || sizeof(R) == 4) {
        div6_vec<N>(v);
    } else if constexpr (MODE == 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            emin = min(emin, __builtin_amdgcn_frexp_exp(v[i]));
            v[i] = div6_fast(v[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double xs = v[i];
            asm volatile("" : "+v"(xs));
            v[i] = xs / 6.0;
        }
    }
}
#ifdef DPILQR_DIV6_PER_SUBSTEP
#define DPILQR_ONE_GUARD(R) false
#else
#define DPILQR_ONE_GUARD(R) (sizeof(R) == 8)
#endif
// run(mode, emin) is a whole RK4 step: once on the fast form, again with divisions if a sub-normal quotient was met
template <typename R, typename F>
__device__ __forceinline__ void rk4_one_guard(F&& run) {
    int emin = 0;
    if constexpr (DPILQR_ONE_GUARD(R)) {
        run(std::integral_constant<int, 1>{}, emin);
        if (__builtin_amdgcn_ballot_w64(emin <= -1000) != 0ull) run(std::integral_constant<int, 2>{}, emin);
    } else {
        run(std::integral_constant<int, 0>{}, emin);
    }
}

// Models whose only transcendental argument is a heading angle with a CONTROL as its rate (Unicycle4D, Car3D: theta' = u[1],
// held over the step): ModelDef<M>::kHeading / f_sc.
template <int M> struct HasHeading { static constexpr bool value = false; };
template <> struct HasHeading<kCar3D> { static constexpr bool value = true; };
template <> struct HasHeading<kUnicycle4D> { static constexpr bool value = true; };

// Models whose transcendental arguments are controls only (Quadcopter6D: tan of two; Human6D and its padded form: sin, cos of one)
template <int M> struct HasControlTrig { static constexpr bool value = false; };
template <> struct HasControlTrig<kQuadcopter6D> { static constexpr bool value = true; };
template <> struct HasControlTrig<kHuman6D> { static constexpr bool value = true; };
template <> struct HasControlTrig<kHumanPad12D> { static constexpr bool value = true; };

// classical RK4 with 5 fixed sub-steps, zero-order-hold u (bbdynamics.cpp:39-93)
template <int M, typename R>
__device__ inline void integrate(const R* x, const R* u, R dt, R* xn) {
    using D = ModelDef<M>;
    constexpr int NS = D::NS;
    const R dh = dt / 5;
    if constexpr (HasHeading<M>::value) {
        // The heading's slope is the same control in every stage (k0 = k1 = k2 = k3 = u[1] in that component): the heading is
        // AFFINE in time over the step.  So (i) the second and the third stage of a sub-step are evaluated at bitwise the same
        // heading, xa + (dh / 2) u[1], and one sincos serves both (15 instead of 20 per step, every result bit for bit what 20
        // give: -DDPILQR_TRIG_DIRECT builds); and (ii), the default, the eleven headings of a step -- theta_0 + j (dh / 2) u[1],
        // j = 0 .. 10 -- are rotations of one another: sincos(theta_0) and sincos((dh / 2) u[1]) once per step, then ten
        // rotations (four multiplications and two additions each).  The rotated values sit a few ulp from the direct ones
        // (each rotation adds one rounding of an O(1) quantity: <= 1e-15 after ten), i.e. inside the difference between this
        // device's sincos and the reference's libm, and two orders of magnitude inside the 1e-12 at which the models are held
        // to the reference's own numbers (G1); the whole GPU suite -- reference golden decision traces of the unicycle solves
        // included -- passes on either build.  Why: the sincos ARE a unicycle's line search, 55-60 % of its launch time
        // (profiles/r04_trig.txt): five unicycles' line search 28.9 -> 23.8 ms per 2048-item solve with (i), -> 13.9 ms with (ii).
        constexpr int H = D::kHeading;
        rk4_one_guard<R>([&](auto mode_tag, int& emin) {
        constexpr int MODE = decltype(mode_tag)::value;
        R k0[NS], k1[NS], k2[NS], k3[NS], xa[NS], xb[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) xn[i] = x[i];
#ifndef DPILQR_TRIG_DIRECT
        R s0, c0, sh, ch;
        sincos_r(xn[H], &s0, &c0);
        sincos_r((dh / R(2.0)) * u[1], &sh, &ch);
#endif
        for (int s = 0; s < 5; ++s) {
            R sn, cs;
#pragma unroll
            for (int i = 0; i < NS; ++i) xa[i] = xn[i];
#ifndef DPILQR_TRIG_DIRECT
            D::f_sc(xa, u, s0, c0, k0);
            sn = s0 * ch + c0 * sh; cs = c0 * ch - s0 * sh;
#pragma unroll
            for (int i = 0; i < NS; ++i) xb[i] = xa[i] + (dh / R(2.0)) * k0[i];
            D::f_sc(xb, u, sn, cs, k1);
#pragma unroll
            for (int i = 0; i < NS; ++i) xb[i] = xa[i] + (dh / R(2.0)) * k1[i];
            D::f_sc(xb, u, sn, cs, k2);
#pragma unroll
            for (int i = 0; i < NS; ++i) xb[i] = xa[i] + dh * k2[i];
            s0 = sn * ch + cs * sh; c0 = cs * ch - sn * sh;
            D::f_sc(xb, u, s0, c0, k3);
#pragma unroll
            for (int i = 0; i < NS; ++i) xb[i] = dh * (k0[i] + R(2.0) * k1[i] + R(2.0) * k2[i] + k3[i]);
            div6_sel<MODE, NS>(xb, emin);
#pragma unroll
            for (int i = 0; i < NS; ++i) xn[i] += xb[i];
            continue;
#endif
            sincos_r(xa[H], &sn, &cs);
            D::f_sc(xa, u, sn, cs, k0);
#pragma unroll
            for (int i = 0; i < NS; ++i) xb[i] = xa[i] + (dh / R(2.0)) * k0[i];
            sincos_r(xb[H], &sn, &cs);
            D::f_sc(xb, u, sn, cs, k1);
#pragma unroll
            for (int i = 0; i < NS; ++i) xb[i] = xa[i] + (dh / R(2.0)) * k1[i];
            D::f_sc(xb, u, sn, cs, k2);          // xb[H] is the previous stage's heading, bit for bit
#pragma unroll
            for (int i = 0; i < NS; ++i) xb[i] = xa[i] + dh * k2[i];
            sincos_r(xb[H], &sn, &cs);
            D::f_sc(xb, u, sn, cs, k3);
#pragma unroll
            for (int i = 0; i < NS; ++i) xb[i] = dh * (k0[i] + R(2.0) * k1[i] + R(2.0) * k2[i] + k3[i]);
            div6_sel<MODE, NS>(xb, emin);
#pragma unroll
            for (int i = 0; i < NS; ++i) xn[i] += xb[i];
        }
        });
        return;
    }
    if constexpr (HasControlTrig<M>::value) {
        // tan / sin / cos of CONTROLS: the same numbers in all twenty stage evaluations of the step, evaluated once (bit-identical
        // results; the inlined library functions branch, and the optimiser neither hoists them out of the sub-step loop nor
        // merges the four of a sub-step)
        R tr[2];
        D::pre(u, tr);
        rk4_one_guard<R>([&](auto mode_tag, int& emin) {
        constexpr int MODE = decltype(mode_tag)::value;
        R k0[NS], k1[NS], k2[NS], k3[NS], xa[NS], xb[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) xn[i] = x[i];
        for (int s = 0; s < 5; ++s) {
#pragma unroll
            for (int i = 0; i < NS; ++i) xa[i] = xn[i];
            D::f_tr(xa, u, tr, k0);
#pragma unroll
            for (int i = 0; i < NS; ++i) xb[i] = xa[i] + (dh / R(2.0)) * k0[i];
            D::f_tr(xb, u, tr, k1);
#pragma unroll
            for (int i = 0; i < NS; ++i) xb[i] = xa[i] + (dh / R(2.0)) * k1[i];
            D::f_tr(xb, u, tr, k2);
#pragma unroll
            for (int i = 0; i < NS; ++i) xb[i] = xa[i] + dh * k2[i];
            D::f_tr(xb, u, tr, k3);
#pragma unroll
            for (int i = 0; i < NS; ++i) xb[i] = dh * (k0[i] + R(2.0) * k1[i] + R(2.0) * k2[i] + k3[i]);
            div6_sel<MODE, NS>(xb, emin);
#pragma unroll
            for (int i = 0; i < NS; ++i) xn[i] += xb[i];
        }
        });
        return;
    }
    if constexpr (M == kDoubleInt4D && sizeof(R) == 8) {
#ifndef DPILQR_DIV6_PER_SUBSTEP   // (A/B builds: one guard per sub-step, the form of rounds 2-4a)
        // The double integrator (cfg2's model): ONE sub-normal guard for the step's twenty divisions by 6 instead of one per
        // sub-step.  The five sub-steps run with the reciprocal form (correctly rounded for every normal quotient) while the
        // smallest binary exponent among the operands is tracked; only if some lane met an operand small enough for a sub-normal
        // quotient is the whole step redone with the division (correctly rounded everywhere).  Either way every quotient is
        // RN(v / 6): the results are those of a guard per sub-step, bit for bit; the fast path loses four ballots / branches
        // and the register copies that the rare branch's merges cost it (72 v_mov_b64 per step).
        auto rk4 = [&](auto exact_tag, int& emin) {
            constexpr bool EXACT = decltype(exact_tag)::value;
            double k0[NS], k1[NS], k2[NS], k3[NS], xa[NS], xb[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) xn[i] = x[i];
#pragma unroll
            for (int s = 0; s < 5; ++s) {
#pragma unroll
                for (int i = 0; i < NS; ++i) xa[i] = xn[i];
                D::f(xa, u, k0);
#pragma unroll
                for (int i = 0; i < NS; ++i) xb[i] = xa[i] + (dh / R(2.0)) * k0[i];
                D::f(xb, u, k1);
#pragma unroll
                for (int i = 0; i < NS; ++i) xb[i] = xa[i] + (dh / R(2.0)) * k1[i];
                D::f(xb, u, k2);
#pragma unroll
                for (int i = 0; i < NS; ++i) xb[i] = xa[i] + dh * k2[i];
                D::f(xb, u, k3);
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    double v = dh * (k0[i] + R(2.0) * k1[i] + R(2.0) * k2[i] + k3[i]);
                    if constexpr (EXACT) {
                        asm volatile("" : "+v"(v));
                        v = v / 6.0;
                    } else {
                        emin = min(emin, __builtin_amdgcn_frexp_exp(v));
                        v = div6_fast(v);
                    }
                    xn[i] += v;
                }
            }
        };
        int emin = 0;
        rk4(std::false_type{}, emin);
        if (__builtin_amdgcn_ballot_w64(emin <= -1000) != 0ull) rk4(std::true_type{}, emin);
        return;
#endif
    }
    constexpr int kSubUnroll = (M == kDoubleInt4D) ? 5 : 1;
    R k0[NS], k1[NS], k2[NS], k3[NS], xa[NS], xb[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) xn[i] = x[i];
    // (the double integrator's five sub-steps unrolled: a dozen instructions each -- line search -1..2 %, rollout -7 %; the models
    // with trigonometry keep the loop: their stage evaluations are hundreds of instructions)
#pragma unroll kSubUnroll
    for (int s = 0; s < 5; ++s) {
#pragma unroll
        for (int i = 0; i < NS; ++i) xa[i] = xn[i];
        D::f(xa, u, k0);
#pragma unroll
        for (int i = 0; i < NS; ++i) xb[i] = xa[i] + (dh / R(2.0)) * k0[i];
        D::f(xb, u, k1);
#pragma unroll
        for (int i = 0; i < NS; ++i) xb[i] = xa[i] + (dh / R(2.0)) * k1[i];
        D::f(xb, u, k2);
#pragma unroll
        for (int i = 0; i < NS; ++i) xb[i] = xa[i] + dh * k2[i];
        D::f(xb, u, k3);
#pragma unroll
        for (int i = 0; i < NS; ++i) xb[i] = dh * (k0[i] + R(2.0) * k1[i] + R(2.0) * k2[i] + k3[i]);
        div6_vec<NS>(xb);
#pragma unroll
        for (int i = 0; i < NS; ++i) xn[i] += xb[i];
    }
}

// forward-Euler discretised Jacobians: A = I + dt*A_c, B = dt*B_c (cpp:95-106)
template <int M, typename R>
__device__ inline void linearize(const R* x, const R* u, R dt, R* A, R* B) {
    using D = ModelDef<M>;
    constexpr int NS = D::NS, NC = D::NC;
#pragma unroll
    for (int i = 0; i < NS * NS; ++i) A[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NS * NC; ++i) B[i] = 0.0;
    D::jac(x, u, A, B);
#pragma unroll
    for (int i = 0; i < NS * NS; ++i) {
        A[i] *= dt;
        if (i % (NS + 1) == 0) A[i] += R(1.0);
    }
#pragma unroll
    for (int i = 0; i < NS * NC; ++i) B[i] *= dt;
}

// ---- runtime dispatch restricted to the models of one (NS,NC) family; arrays sized by the family
template <int NS> struct Family;
#define DPILQR_FAMILY(NSV, ...)                                                                          \
    template <> struct Family<NSV> {                                                                     \
        template <typename R> __device__ static void f(int model, const R* x, const R* u, R* o) {        \
            switch (model) { __VA_ARGS__(DPILQR_CASE_F) default: break; }                                \
        }                                                                                                \
        template <typename R> __device__ static void integrate(int model, const R* x, const R* u, R dt, R* xn) { \
            switch (model) { __VA_ARGS__(DPILQR_CASE_I) default: break; }                                \
        }                                                                                                \
        template <typename R> __device__ static void linearize(int model, const R* x, const R* u, R dt, R* A, R* B) { \
            switch (model) { __VA_ARGS__(DPILQR_CASE_L) default: break; }                                \
        }                                                                                                \
    };
#define DPILQR_CASE_F(M) case M: ModelDef<M>::f(x, u, o); break;
#define DPILQR_CASE_I(M) case M: ::dpilqr::integrate<M>(x, u, dt, xn); break;
#define DPILQR_CASE_L(M) case M: ::dpilqr::linearize<M>(x, u, dt, A, B); break;
#define DPILQR_FAM3(X) X(kCar3D)
#define DPILQR_FAM4(X) X(kDoubleInt4D) X(kUnicycle4D)
#define DPILQR_FAM6(X) X(kDoubleInt6D) X(kQuadcopter6D) X(kHuman6D) X(kHumanLin6D)
#define DPILQR_FAM12(X) X(kQuadcopter12D) X(kHumanPad12D)
DPILQR_FAMILY(3, DPILQR_FAM3)
DPILQR_FAMILY(4, DPILQR_FAM4)
DPILQR_FAMILY(6, DPILQR_FAM6)
DPILQR_FAMILY(12, DPILQR_FAM12)

template <int NS, typename R>
__device__ inline void f_rt(int model, const R* x, const R* u, R* o) { Family<NS>::f(model, x, u, o); }
template <int NS, typename R>
__device__ inline void integrate_rt(int model, const R* x, const R* u, R dt, R* xn) { Family<NS>::integrate(model, x, u, dt, xn); }
template <int NS, typename R>
__device__ inline void linearize_rt(int model, const R* x, const R* u, R dt, R* A, R* B) { Family<NS>::linearize(model, x, u, dt, A, B); }

}  // namespace dpilqr
