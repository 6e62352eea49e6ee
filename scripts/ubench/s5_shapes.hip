// micro-benchmark (round 6): the fused cfg2 sweep's S5-shaped product D = X^T Y, X and Y 12 reduction rows in LDS, D 20 x 21
// (a1 = T3 [K|d] at n_x = 20, n_u = 10), formed (A) as today: 2 x 2 tiles of v_mfma_f64_16x16x4_f64, three reduction steps -- tiles
// 35 - 41 % full -- or (B) on v_mfma_f64_4x4x4_4b_f64: 5 x 6 blocks of 4 x 4, four blocks per instruction, three reduction steps --
// blocks 100 % full in rows, 21 of 24 columns.  Same operand strides as the kernel (62 doubles).  Cycles per product per wavefront at
// 1, 2, 3 wavefronts per SIMD: is the 4x4 form's higher instruction count (24 + 48 LDS reads against 12 + 12) paid back by its
// 21-cycle issue?    hipcc --offload-arch=gfx950 -O3 -o s5_shapes scripts/ubench/s5_shapes.hip && ./s5_shapes
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int LD = 62, ROWS = 12;

template <int FORM>
__global__ void k_prod(double* out, int iters) {
    __shared__ double sX[4][ROWS * LD], sY[4][ROWS * LD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* X = sX[wave & 3]; double* Y = sY[wave & 3];
    for (int e = lane; e < ROWS * LD; e += 64) { X[e] = 1e-3 * (e % 17) + 0.5; Y[e] = 1e-3 * (e % 13) - 0.25; }
    __syncthreads();
    double sink = 0.0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (FORM == 0) {
        const int g = lane >> 4, c = lane & 15;
        for (int it_ = 0; it_ < iters; ++it_) {
            v4d d[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                double a[2], b[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) { a[q] = X[(4 * ks + g) * LD + 16 * q + c]; b[q] = Y[(4 * ks + g) * LD + 16 * q + c]; }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) d[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], d[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) sink += d[i][j][0] + d[i][j][1] + d[i][j][2] + d[i][j][3];
            X[lane] = sink * 1e-30 + 0.5;      // (a dependence between iterations, as a step's result feeds the next step)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else {
        // block slot s = lane / 16 of accumulator q holds D-block number 4 q + s = bi * 6 + bj (30 blocks, the last two slots idle)
        const int s = lane >> 4, li = lane & 3, lk = (lane >> 2) & 3;
        int xo[8], yo[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int blk = min(4 * q + s, 29), bi = blk / 6, bj = blk - 6 * bi;
            xo[q] = lk * LD + 4 * bi + li; yo[q] = lk * LD + 4 * bj + li;
        }
        for (int it_ = 0; it_ < iters; ++it_) {
            double d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < 3; ++kb)
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    d[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(X[4 * kb * LD + xo[q]], Y[4 * kb * LD + yo[q]], d[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) sink += d[q];
            X[lane] = sink * 1e-30 + 0.5;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0) / iters;
}

template <int FORM>
static void run(double* out, const char* name) {
    for (int wps = 1; wps <= 3; ++wps) {
        const int threads = 256 * wps;      // wps wavefronts per SIMD, one workgroup per CU
        hipLaunchKernelGGL(k_prod<FORM>, dim3(256), dim3(threads), 0, 0, out, 200);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_prod<FORM>, dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double clk; hipMemcpy(&clk, out + (1 << 20), 8, hipMemcpyDeviceToHost);
        printf("%-34s %d wavefront(s) per SIMD: %7.1f shader clocks per product per wavefront in-kernel, %7.1f ns per product per SIMD (wall)\n",
               name, wps, clk, ms * 1e6 / iters / wps);
    }
}

int main() {
    double* out; hipMalloc(&out, 8 * ((1 << 20) + 8));
    run<0>(out, "(A) 16x16x4, 2x2 tiles x 3 steps");
    run<1>(out, "(B) 4x4x4_4b, 8 accumulators x 3");
    return 0;
}
