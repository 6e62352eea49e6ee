// dpilqr_amd/csrc/trig_inline.hpp against the device library, bit for bit: sincos() and tan() of 2^26 fp64 arguments per range
// (uniform in [-4, 4]; log-uniform magnitudes 1e-300 .. 2^30 of either sign; multiples of pi/2 plus or minus a few ulps).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Idpilqr_amd/csrc -o scripts/ubench/trig_inline_check scripts/ubench/trig_inline_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "trig_inline.hpp"

__device__ inline uint64_t mix(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

__global__ void k(int range, unsigned long long n, unsigned long long* bad, double* first_bad) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = mix(i * 3 + range), r2 = mix(r);
    const double u = (double)(r >> 11) * 0x1.0p-53, u2 = (double)(r2 >> 11) * 0x1.0p-53;
    double x;
    if (range == 0) x = 8.0 * u - 4.0;
    else if (range == 1) x = ((r2 & 1) ? -1.0 : 1.0) * exp2(-996.0 + u * (996.0 + 30.0) - 1e-9);
    else { const double kq = floor(u * 4096.0) * 1.5707963267948966; x = __longlong_as_double(__double_as_longlong(kq) + (long long)(r2 % 9) - 4); if (u2 < 0.5) x = -x; }
    if (!(fabs(x) < 0x1.0p+30)) return;
    double s0, c0, s1, c1;
    sincos(x, &s0, &c0);
    const double t0 = tan(x);
    const dpilqr::TrigRed rr = dpilqr::trig_reduce(x);
    dpilqr::trig_sincos(x, rr, &s1, &c1);
    const double t1 = dpilqr::trig_tan(x, rr);
    const bool same = __double_as_longlong(s0) == __double_as_longlong(s1) && __double_as_longlong(c0) == __double_as_longlong(c1) &&
                      __double_as_longlong(t0) == __double_as_longlong(t1);
    if (!same) { if (atomicAdd(bad, 1ull) == 0ull) { first_bad[0] = x; first_bad[1] = s0; first_bad[2] = s1; first_bad[3] = c0; first_bad[4] = c1; first_bad[5] = t0; first_bad[6] = t1; } }
}

int main() {
    unsigned long long* bad; double* fb;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&fb, 7 * 8);
    const unsigned long long n = 1ull << 26;
    const char* names[3] = {"uniform in [-4, 4]", "log-uniform magnitudes 1e-300 .. 2^30", "multiples of pi/2 +- 4 ulps"};
    int rc = 0;
    for (int range = 0; range < 3; ++range) {
        (void)hipMemset(bad, 0, 8);
        hipLaunchKernelGGL(k, dim3((unsigned)(n / 256)), dim3(256), 0, 0, range, n, bad, fb);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        unsigned long long b; double f[7];
        (void)hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(f, fb, 56, hipMemcpyDeviceToHost);
        printf("%-42s %llu arguments: %llu differ from sincos() / tan()", names[range], n, b);
        if (b) { printf("  first: x %.17g sin %.17g / %.17g cos %.17g / %.17g tan %.17g / %.17g", f[0], f[1], f[2], f[3], f[4], f[5], f[6]); rc = 1; }
        printf("\n");
    }
    return rc;
}
