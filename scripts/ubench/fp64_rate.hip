// fp64 vector issue rate per wavefront on one CU of gfx950, with 1, 2, 3, 4 and 8 wavefronts in the workgroup (one per SIMD up to four):
// does a wavefront's v_fma_f64 rate drop when the other SIMDs of its CU run fp64 too?
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/fp64_rate scripts/ubench/fp64_rate.hip && scripts/ubench/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DEP>
__global__ void k(double* out, long long* clk, int n) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + threadIdx.x * 1e-6 + i;
    const double b = 1.0000001, c = 1e-9;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        if (DEP) { for (int i = 0; i < 8; ++i) a[0] = __builtin_fma(a[0], b, c); }      // one dependent chain
        else { for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], b, c); }          // eight independent ones
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[threadIdx.x >> 6] = t1 - t0;
}
int main() {
    double* out; long long* clk;
    (void)hipMalloc(&out, 1024 * 8); (void)hipMalloc(&clk, 16 * 8);
    const int n = 20000;
    for (int dep = 0; dep < 2; ++dep)
        for (int w : {1, 2, 3, 4, 8}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (dep) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * w), 0, 0, out, clk, n);
                else hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * w), 0, 0, out, clk, n);
                (void)hipDeviceSynchronize();
            }
            long long c[16]; (void)hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
            printf("%s, %d wavefront(s): %.2f clocks per v_fma_f64 (wavefront 0)\n", dep ? "one dependent chain " : "eight independent   ", w, (double)c[0] / n / 8);
        }
    return 0;
}
