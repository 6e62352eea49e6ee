// Device check: div6(x) (models.hpp) against the fp64 division on random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "models.hpp"
__global__ void k(uint64_t seed, unsigned long long* bad, double* ex) {
    uint64_t s = seed + 0x9E3779B97F4A7C15ULL * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (int it = 0; it < 4096; ++it) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint64_t b = s;
        if (it & 1) b = (b & 0x800FFFFFFFFFFFFFULL) | ((uint64_t)(1023 - 60 + (s >> 40) % 120) << 52);
        double x = __longlong_as_double((long long)b);
        if (x != x || isinf(x)) continue;
        const double a = dpilqr::div6(x), t = x / 6.0;
        if (a != t) { if (atomicAdd(bad, 1ULL) == 0) { ex[0] = x; ex[1] = a; ex[2] = t; } }
    }
}
int main() {
    unsigned long long* bad; double* ex;
    hipMalloc(&bad, 8); hipMalloc(&ex, 24); hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, 12345ULL, bad, ex);
    unsigned long long h; double hx[3];
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hx, ex, 24, hipMemcpyDeviceToHost);
    printf("mismatches %llu of %llu", h, 4096ULL * 256 * 4096);
    if (h) printf("  first: x=%a div6=%a div=%a", hx[0], hx[1], hx[2]);
    printf("\n");
    return 0;
}
