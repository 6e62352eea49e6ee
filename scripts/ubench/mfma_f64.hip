// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and its lane layout (diagnostic, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_rate(double* out, int iters, int chains) {
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        if (chains > 1) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        if (chains > 2) c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        if (chains > 3) c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    v4d s = c0 + c1 + c2 + c3;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (iters * chains);
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
}

__global__ void k_layout(const double* A, const double* B, double* D) {
    // A is 16x4 (row-major), B is 4x16: find out which element each lane must supply / receives
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + (l / 16)];   // guess: lane -> A[i = l%16][k = l/16]
    const double b = B[(l / 16) * 16 + (l % 16)];  // guess: lane -> B[k = l/16][j = l%16]
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
}

int main() {
    double* out; hipMalloc(&out, sizeof(double) * (1 + 1024 * 1024));
    for (int chains = 1; chains <= 4; ++chains)
        for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
            const int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves, one per SIMD
            hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, out, 2000, chains);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, out, 20000, chains);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double cyc; hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost);
            const double n_mfma_per_simd = 20000.0 * chains * waves_per_simd;
            printf("chains %d waves/SIMD %d: in-wave %.1f memtime ticks per MFMA; wall %.3f ms -> %.1f ns per MFMA per SIMD (%.1f cycles @2.4GHz), %.1f TFLOP/s chip\n",
                   chains, waves_per_simd, cyc, ms, ms * 1e6 / n_mfma_per_simd, ms * 1e6 / n_mfma_per_simd * 2.4,
                   n_mfma_per_simd * 1024 * 2048.0 / (ms * 1e-3) / 1e12);
        }
    // layout check
    std::vector<double> A(64), B(64), Dh(256), ref(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1 + i; B[i] = 100 + 3 * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) ref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(Dh.data(), dD, 2048, hipMemcpyDeviceToHost);
    // hypothesis: lane l, v -> D[i = 4*(l/16) + v][j = l%16]
    int ok1 = 1, ok2 = 1;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        if (Dh[l * 4 + v] != ref[(4 * (l / 16) + v) * 16 + (l % 16)]) ok1 = 0;
        if (Dh[l * 4 + v] != ref[((l / 16) + 4 * v) * 16 + (l % 16)]) ok2 = 0;
    }
    printf("layout A[i=l%%16][k=l/16], B[k=l/16][j=l%%16]: D[i=4*(l/16)+v][j=l%%16] %s ; D[i=(l/16)+4v][j=l%%16] %s\n", ok1 ? "YES" : "no", ok2 ? "YES" : "no");
    return 0;
}
