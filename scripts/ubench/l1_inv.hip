// Does `buffer_inv sc0` (workgroup scope) drop a CU's vector L1 on gfx950 outside threadgroup-split mode?
// Two workgroups on ONE XCD (blocks 0 and 8 of a 16-block launch; both report HW_REG_XCC_ID): A reads an array with plain loads
// (it now sits in A's L1), B overwrites it and waits for its stores (s_waitcnt vmcnt(0): they are in the shared L2), A then
// invalidates in one of four ways and reads again with plain loads.  Stale words = words still holding the old value.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/l1_inv scripts/ubench/l1_inv.hip && scripts/ubench/l1_inv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ int ld_plain(const int* p) {
    int v;
    asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ int ld_sc0(const int* p) {
    int v;
    asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ int ld_sc1(const int* p) {
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ int ld_nt(const int* p) {
    int v;
    asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ int poll(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// words: the array (n ints); ctl[0]: A has read round r; ctl[1]: B has written round r; out[mode*? ...]
__global__ void k(int* words, int n, int* ctl, int* stale, int* xcc, int mode, int rounds) {
    const int blk = blockIdx.x;
    if (blk != 0 && blk != 8) return;
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) xcc[blk ? 1 : 0] = x & 15;
    int bad = 0;
    unsigned long long clocks = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (blk == 0) {       // A
            // read: the previous round's values into L1
            for (int i = threadIdx.x; i < n; i += blockDim.x) (void)ld_plain(words + i);
            __syncthreads();
            if (threadIdx.x == 0) { __hip_atomic_store(&ctl[0], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); while (poll(&ctl[1]) < r) __builtin_amdgcn_s_sleep(2); }
            __syncthreads();
            if (threadIdx.x < 64) {
                if (mode == 1) asm volatile("buffer_inv sc0" ::: "memory");
                if (mode == 2) asm volatile("buffer_inv sc1" ::: "memory");
                if (mode == 3) asm volatile("buffer_inv sc0 sc1" ::: "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            if (mode == 4) { for (int i = threadIdx.x; i < n; i += blockDim.x) bad += ld_sc0(words + i) != r; }
            else if (mode == 5) { for (int i = threadIdx.x; i < n; i += blockDim.x) bad += ld_sc1(words + i) != r; }
            else if (mode == 6) { for (int i = threadIdx.x; i < n; i += blockDim.x) bad += ld_nt(words + i) != r; }
            else { for (int i = threadIdx.x; i < n; i += blockDim.x) bad += ld_plain(words + i) != r; }
            if (threadIdx.x == 0) clocks += __builtin_amdgcn_s_memtime() - t0;
        } else {              // B
            if (threadIdx.x == 0) while (poll(&ctl[0]) < r) __builtin_amdgcn_s_sleep(2);
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += blockDim.x) words[i] = r;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stores are in the XCD's L2; NO write-back of it
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(&ctl[1], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (blk == 0) { atomicAdd(stale, bad); if (threadIdx.x == 0) xcc[2] = (int)(clocks / rounds); }
}

int main() {
    const int n = 2048, rounds = 200;      // 8 KB: well inside a 32 KB L1
    int *words, *ctl, *stale, *xcc;
    hipMalloc(&words, n * 4); hipMalloc(&ctl, 8); hipMalloc(&stale, 4); hipMalloc(&xcc, 12);
    const char* names[7] = {"no invalidate", "buffer_inv sc0 (workgroup)", "buffer_inv sc1 (agent)", "buffer_inv sc0 sc1 (system)",
                            "no invalidate, loads sc0", "no invalidate, loads sc1", "no invalidate, loads nt"};
    for (int mode = 0; mode < 7; ++mode) {
        hipMemset(words, 0, n * 4); hipMemset(ctl, 0, 8); hipMemset(stale, 0, 4);
        hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, 0, words, n, ctl, stale, xcc, mode, rounds);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        int s, xc[3];
        hipMemcpy(&s, stale, 4, hipMemcpyDeviceToHost); hipMemcpy(xc, xcc, 12, hipMemcpyDeviceToHost);
        printf("%-30s XCDs %d %d: %d stale words of %d; second read of %d words: %d clocks (8 dependent loads per lane)\n", names[mode], xc[0], xc[1], s, n * rounds, n, xc[2]);
    }
    return 0;
}
