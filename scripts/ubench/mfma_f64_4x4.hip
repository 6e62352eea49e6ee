// micro-benchmark: issue rate and lane layout of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per
// instruction; diagnostic, not product code).  The layout is found by probing with one-hot operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_rate(double* out, int iters, int chains) {
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        if (chains > 1) c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        if (chains > 2) c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        if (chains > 3) c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        if (chains > 4) {
            c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (iters * chains);
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

// one-hot probe: lane la supplies a = 1 (others 0), lane lb supplies b = 1 (others 0); D[l] for all lanes
__global__ void k_probe(double* D) {
    const int l = threadIdx.x, la = blockIdx.x / 64, lb = blockIdx.x % 64;
    const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
    D[blockIdx.x * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}

int main() {
    double* out; hipMalloc(&out, sizeof(double) * (1 + 1024 * 1024));
    for (int chains : {1, 2, 4, 8})
        for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
            const int blocks = 256 * waves_per_simd;
            hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, out, 2000, chains);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, out, 20000, chains);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = 20000.0 * chains * waves_per_simd;
            printf("4x4x4_4b chains %d waves/SIMD %d: %.1f ns per MFMA per SIMD (%.1f cycles @2.4GHz)\n", chains, waves_per_simd,
                   ms * 1e6 / n, ms * 1e6 / n * 2.4);
        }
    double* dD; hipMalloc(&dD, 8 * 64 * 64 * 64);
    hipLaunchKernelGGL(k_probe, dim3(64 * 64), dim3(64), 0, 0, dD);
    std::vector<double> D(64 * 64 * 64);
    hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
    // for every output lane: which (la, lb) pairs contribute
    for (int l : {0, 1, 4, 5, 16, 21, 63}) {
        printf("D lane %2d <- sum over (A lane, B lane):", l);
        for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (D[(la * 64 + lb) * 64 + l] != 0.0) printf(" (%d,%d)", la, lb);
        printf("\n");
    }
    // hypothesis: block = l/16; A lane (b, i = l%4, k = (l/4)%4); B lane (b, j = l%4, k = (l/4)%4); D lane (b, j = l%4, i = (l/4)%4)
    int ok = 1, ok2 = 1;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) for (int l = 0; l < 64; ++l) {
        const double got = D[(la * 64 + lb) * 64 + l];
        const int ba = la / 16, ia = la % 4, ka = (la / 4) % 4, bb = lb / 16, jb = lb % 4, kb = (lb / 4) % 4;
        const double h1 = (ba == bb && ka == kb && l / 16 == ba && l % 4 == jb && (l / 4) % 4 == ia) ? 1.0 : 0.0;
        const double h2 = (ba == bb && ka == kb && l / 16 == ba && l % 4 == ia && (l / 4) % 4 == jb) ? 1.0 : 0.0;
        if (got != h1) ok = 0;
        if (got != h2) ok2 = 0;
    }
    printf("layout A(b=l/16,i=l%%4,k=(l/4)%%4) B(b,j=l%%4,k=(l/4)%%4): D(b, j=l%%4, i=(l/4)%%4) %s ; D(b, i=l%%4, j=(l/4)%%4) %s\n", ok ? "YES" : "no", ok2 ? "YES" : "no");
    return 0;
}
