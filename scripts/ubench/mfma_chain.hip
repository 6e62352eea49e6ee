// Issue-to-issue time of DEPENDENT v_mfma_f64_16x16x4_f64 on gfx950 (one wavefront alone on its SIMD): one chain, two interleaved
// chains, four.  What a reduction that must add its terms in order (K[t] dx of the line search, forward.hpp) pays per step.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/mfma_chain scripts/ubench/mfma_chain.hip && scripts/ubench/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void k(double* out, long long* clk, int n) {
    v4d acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = v4d{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *clk = __builtin_amdgcn_s_memtime() - t0;
}
int main() {
    double* out; long long* clk;
    hipMalloc(&out, 64 * 8); hipMalloc(&clk, 8);
    const int n = 4096;
    long long c;
#define RUN(CH) hipLaunchKernelGGL(k<CH>, dim3(1), dim3(64), 0, 0, out, clk, n); hipDeviceSynchronize(); hipLaunchKernelGGL(k<CH>, dim3(1), dim3(64), 0, 0, out, clk, n); hipDeviceSynchronize(); \
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost); printf("%d chain(s): %.1f clocks per product (%.1f per step of %d)\n", CH, (double)c / n / CH, (double)c / n, CH);
    RUN(1) RUN(2) RUN(4)
    return 0;
}
