// Straight-line sin / cos / tan of fp64 arguments below 2^30 in magnitude: the device library's own algorithm, operation for
// operation, without its branch.
//
// Why: sincos() and tan() of the ROCm device library (ocml) choose between two argument reductions by a branch (|x| < 2^30: three
// fused multiply-adds by the parts of pi/2; else Payne-Hanek).  Inlined, every call is its own little control-flow graph, and the
// compiler can neither interleave the calls of one stage evaluation nor share what they have in common.  A twelve-state
// quadcopter's f() makes four of them -- sincos of three Euler angles and tan of one of those -- twenty times per integration
// step.  Written out as one basic block the four chains interleave and tan(theta) shares theta's reduction with sincos(theta).
// Measured at BASELINE config 5's size (one item; scripts/bench_big.py, scripts/r06_fwd_phases.sh): the bare rollout -- one
// wavefront, latency-bound -- 4.46 -> 3.85 ms; the ten-candidate line search unchanged (69 k of its 135 k clocks per step are the
// integration, but there four wavefronts issue ~6000 fp64 instructions each per step, and a SIMD's fp64 issue rate, not the
// chains' latency, is what bounds them).
//
// What: `trig_reduce`, `trig_sincos`, `trig_tan` below ARE ocml's small-argument path -- __ocmlpriv_trigredsmall_f64,
// __ocmlpriv_sincosred2_f64, __ocmlpriv_tanred2_f64 and the sign logic of __ocml_sincos_f64 / __ocml_tan_f64 as this image's
// /opt/rocm/amdgcn/bitcode/ocml.bc (ROCm 7.2.0) defines them, the same operations in the same order with the same constants, so
// the results are the library's bit for bit (scripts/ubench/trig_inline_check.hip: 2^26 arguments per function, none different;
// this translation unit is compiled with -ffp-contract=off, so a product and a sum written apart stay apart).  The caller asks
// `trig_small_all()` first -- one wave-wide test -- and calls the library where an argument is 2^30 or larger, infinite or NaN.
#pragma once
#include <hip/hip_runtime.h>

namespace dpilqr {

struct TrigRed { double hi, lo; int q; };      // x = q (pi / 2) + hi + lo, q mod 4

// every lane's |x| is below the reduction's limit (false for infinities and NaNs)
__device__ __forceinline__ bool trig_small_all(double a, double b, double c) {
    const double lim = 0x1.0p+30;
    return __builtin_amdgcn_ballot_w64(!(fabs(a) < lim && fabs(b) < lim && fabs(c) < lim)) == 0ull;
}

// __ocmlpriv_trigredsmall_f64 of |x|
__device__ __forceinline__ TrigRed trig_reduce(double x) {
    const double ax = fabs(x);
    const double t2 = ax * 0x1.45f306dc9c883p-1;                 // 2 / pi
    const double dn = __builtin_rint(t2);
    const double t4 = fma(dn, -0x1.921fb54442d18p+0, ax);        // - dn (pi/2)_head
    const double t5 = fma(dn, -0x1.1a62633145c00p-54, t4);       // - dn (pi/2)_middle
    const double t6 = dn * 0x1.1a62633145c00p-54;
    const double t8 = fma(dn, 0x1.1a62633145c00p-54, -t6);
    const double t9 = t4 - t6;
    const double t10 = t4 - t9;
    const double t11 = t10 - t6;
    const double t12 = t9 - t5;
    const double t13 = t12 + t11;
    const double t14 = t13 - t8;
    const double t15 = fma(dn, -0x1.b839a252049c0p-104, t14);    // - dn (pi/2)_tail
    const double t16 = t5 + t15;
    const double t17 = t16 - t5;
    const double t18 = t15 - t17;
    TrigRed r;
    r.hi = t16; r.lo = t18; r.q = (int)dn & 3;
    return r;
}

// __ocmlpriv_sincosred2_f64 and the quadrant / sign logic of __ocml_sincos_f64
__device__ __forceinline__ void trig_sincos(double x, const TrigRed& r, double* s_out, double* c_out) {
    const double h = r.hi, l = r.lo;
    const double t3 = h * h;
    const double t4 = t3 * 0.5;
    const double t5 = 1.0 - t4;
    const double t6 = 1.0 - t5;
    const double t7 = t6 - t4;
    const double t8 = t3 * t3;
    double p = fma(t3, -0x1.907db46cc5e42p-37, 0x1.1eeb69037ab78p-29);
    p = fma(t3, p, -0x1.27e4fa17f65f6p-22);
    p = fma(t3, p, 0x1.a01a019f4ec90p-16);
    p = fma(t3, p, -0x1.6c16c16c16967p-10);
    p = fma(t3, p, 0x1.5555555555555p-5);
    const double nl = -l;
    const double t15 = fma(h, nl, t7);
    const double t16 = fma(t8, p, t15);
    const double cs = t5 + t16;                                   // cos(hi + lo)
    double q = fma(t3, 0x1.5e0b2f9a43bb8p-33, -0x1.ae600b42fdfa7p-26);
    q = fma(t3, q, 0x1.71de3796cde01p-19);
    q = fma(t3, q, -0x1.a01a019e83e5cp-13);
    q = fma(t3, q, 0x1.1111111110bb3p-7);
    const double t23 = h * (-t3);
    const double t24 = l * 0.5;
    const double t25 = fma(t23, q, t24);
    const double t26 = fma(t3, t25, nl);
    const double t27 = fma(t23, -0x1.5555555555555p-3, t26);
    const double sn = h - t27;                                    // sin(hi + lo)
    const unsigned flip = r.q > 1 ? 0x80000000u : 0u;
    const bool even = (r.q & 1) == 0;
    unsigned long long sb = __double_as_longlong(even ? sn : cs);
    const unsigned xs = (unsigned)((unsigned long long)__double_as_longlong(x) >> 32) & 0x80000000u;
    sb ^= (unsigned long long)(xs ^ flip) << 32;
    unsigned long long cb = __double_as_longlong(even ? cs : -sn);
    cb ^= (unsigned long long)flip << 32;
    *s_out = __longlong_as_double((long long)sb);
    *c_out = __longlong_as_double((long long)cb);
}

// __ocmlpriv_tanred2_f64 and the sign logic of __ocml_tan_f64
__device__ __forceinline__ double trig_tan(double x, const TrigRed& r) {
    const double h = r.hi, l = r.lo;
    const double t4 = h * h;
    const double t6 = fma(h, h, -t4);
    const double t7 = l * 2.0;
    const double t8 = fma(h, t7, t6);
    const double t9 = t4 + t8;
    double p = fma(t9, 0x1.5e089c751c08cp-16, -0x1.78809a9a29f71p-15);
    p = fma(t9, p, 0x1.7746f90a8aae0p-14);
    p = fma(t9, p, -0x1.bb44da6fbf144p-16);
    p = fma(t9, p, 0x1.1e634a7943acfp-13);
    p = fma(t9, p, 0x1.d250fdeb68febp-13);
    p = fma(t9, p, 0x1.37fd9b58c4d95p-11);
    p = fma(t9, p, 0x1.7d5af15120e2cp-10);
    p = fma(t9, p, 0x1.d6d93e09491dfp-9);
    p = fma(t9, p, 0x1.226e12033784dp-7);
    p = fma(t9, p, 0x1.664f49ac36ae2p-6);
    p = fma(t9, p, 0x1.ba1ba1b451c21p-5);
    p = fma(t9, p, 0x1.11111111185b7p-3);
    p = fma(t9, p, 0x1.55555555554eep-2);
    const double t23 = t9 * p;
    const double t24 = h * t23;
    const double t26 = fma(h, t23, -t24);
    const double t27 = h + t24;
    const double t28 = t27 - h;
    const double t29 = t24 - t28;
    const double t30 = l + t26;
    const double t31 = t30 + t29;
    const double t32 = t27 + t31;                                 // tan(hi + lo), head
    const double t33 = t32 - t27;
    const double t34 = t31 - t33;                                 // ... tail
    double rc = __builtin_amdgcn_rcp(t32);
    const double n32 = -t32;
    const double t37 = fma(n32, rc, 1.0);
    const double t38 = fma(t37, rc, rc);
    const double t39 = fma(n32, t38, 1.0);
    const double t40 = fma(t39, t38, t38);
    const double t41 = t32 * t40;
    const double t43 = fma(t40, t32, -t41);
    const double t44 = fma(t40, t34, t43);
    const double t45 = t41 + t44;
    const double t46 = t45 - t41;
    const double t47 = t44 - t46;
    const double t48 = 1.0 - t45;
    const double t49 = 1.0 - t48;
    const double t50 = t49 - t45;
    const double t51 = t50 - t47;
    const double t52 = t48 + t51;
    const double t53 = t40 * t52;
    const double t54 = t40 + t53;                                 // 1 / tan(hi + lo)
    const double v = (r.q & 1) == 0 ? t32 : -t54;
    const unsigned xs = (unsigned)((unsigned long long)__double_as_longlong(x) >> 32) & 0x80000000u;
    return __longlong_as_double((long long)((unsigned long long)__double_as_longlong(v) ^ ((unsigned long long)xs << 32)));
}

}  // namespace dpilqr
