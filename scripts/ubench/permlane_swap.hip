// Semantics of gfx950's v_permlane32_swap / v_permlane16_swap, observed: every lane starts with its own id in A and 100 + id in B.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/permlane_swap scripts/ubench/permlane_swap.hip && scripts/ubench/permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int a = threadIdx.x, b = 100 + threadIdx.x;
    auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r32[0]; out[64 + threadIdx.x] = r32[1]; out[128 + threadIdx.x] = r16[0]; out[192 + threadIdx.x] = r16[1];
}
int main() {
    int* d; (void)hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane32_swap: first result ", "permlane32_swap: second result", "permlane16_swap: first result ", "permlane16_swap: second result"};
    for (int r = 0; r < 4; ++r) { printf("%s, lanes 0 16 32 48:", names[r]); for (int l = 0; l < 64; l += 16) printf(" %4d", h[64 * r + l]); printf("\n"); }
    return 0;
}
