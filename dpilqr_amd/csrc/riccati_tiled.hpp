// riccati_tiled.hpp -- K2, compile-time-sized: the Riccati backward sweep with ONE WAVEFRONT per
// sub-problem (ilqrSolver._backward_pass, control.py:116-148).
//
// Same recursion as riccati.hpp (see the equations there); this kernel is the fast path for the
// sizes it is instantiated for.  Design, for n = n_x, m = n_u known at compile time:
//
//  * the value function [P | p] (and a copy with mu on the diagonal, for the two products that the
//    reference regularises) stays in LDS for the whole horizon; nothing but the tile records is read
//    from HBM and nothing but K[t], d[t] is written.
//  * each step's record is fetched ONE STEP AHEAD straight into registers (16-byte loads issued at the
//    top of step t for record t-1, consumed at the top of step t-1), so the HBM latency hides under
//    the arithmetic of a whole step; only [A|B] (used by two products) is then parked in LDS.
//  * the dense products run as register-blocked outer products: a lane owns an RB x CB block of the
//    output, reads RB + CB operands per reduction step with ds_read_b128 and issues RB*CB fp64 FMAs.
//    Products are STACKED so that one pass serves several of the reference's expressions:
//        S1  [A|B]^T [P|p]      -> A^T P, B^T (P + mu I), A^T p, B^T p          ((n+m) x (n+1))
//        S2  [T1;T2] [A|B]      -> (A^T P) A, (B^T P~) A, (B^T P~) B           ((n+m) x (n+m))
//        S5  T3 [K|d], K^T [Q_ux|Q_u], Q_ux^T [K|d]   -> P and p updates       (n x (n+1), three sums)
//    Operands are kept in LDS in the orientation that makes every operand read contiguous.
//  * the Q_uu solve is LU with partial pivoting held entirely in registers: lane c owns column c of
//    the augmented matrix [Q_uu | Q_ux | Q_u]; pivot row and multipliers travel by v_readlane.
//  * no workgroup barriers: a single wave executes its LDS operations in order.
#pragma once
#include <hip/hip_runtime.h>

#include "tiles.hpp"

namespace dpilqr {

typedef double v2d __attribute__((ext_vector_type(2)));

#define DPILQR_LDS_FENCE() asm volatile("" ::: "memory")

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// Global stores issued through inline asm, on purpose.  hipcc's wait-count pass treats a vmcnt with both
// loads and stores pending as out of order and answers every wait on a prefetched load with vmcnt(0),
// which would also wait for loads issued a few instructions earlier.  gfx950 retires vector-memory
// operations in issue order (MI355X guide, s_waitcnt notes), so hiding these fire-and-forget stores from
// the pass keeps its counted vmcnt(N) waits: they merely become conservative by the number of stores
// in flight.  Nothing ever reads the stored data back inside the kernel.
// The trailing s_nop covers the "VMEM store data > 64 bit, then a VALU write of the data VGPRs" hazard: the
// compiler's hazard recogniser does not look inside inline assembly, and the data registers are read a few
// cycles after issue.
__device__ __forceinline__ void store_v2d_nt(double* p, v2d v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_f64_nt(double* p, double v) {
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}

// Diagnostic stamps (dpilqr_debug_stamps): when a buffer is registered, lane 0 of every sweep workgroup
// records {start, end} of s_memrealtime (100 MHz) and its XCC / CU / SIMD ids.  Never read by any kernel.
static __device__ unsigned long long* g_stamp_buf = nullptr;   // one copy per translation unit (launch.hpp)

constexpr int round_up(int x, int q) { return (x + q - 1) / q * q; }

template <int N, int M>
struct TiledCfg {
    static constexpr int NM = N + M;
    static constexpr int RB = (N % 4 == 0) ? 4 : 2;   // block edge of the stacked products (divides N)
    static constexpr int CB = RB;
    static constexpr int NMP = round_up(NM, RB);      // padded stacked dimension
    static constexpr int LAB = NMP;                   // leading dim of sAB and sT
    static constexpr int LP = round_up(N + 1, CB);    // [P | p | pad]
    static constexpr int NP = round_up(N + 1, 2);     // [K | d | pad], [Q_xx | Q_x | pad] logical width
    static constexpr int LQ = NP;
    static constexpr int LK = NP;
    static constexpr int LG = round_up(M + N + 2, 4); // [Q_uu | Q_ux | Q_u | pad]
    // LDS carve (doubles).  Lifetimes within a step: sAB S0-S2, sT S1-S2, sQ S1-S6, sG S1-S5, sK S3-S6,
    // sT3 S4-S5, sP S6-S1(next).  sK and sT3 are therefore carved out of sT's space (dead after S2), which brings a
    // wave's slice to ~20 KB so that EIGHT waves (two per SIMD) fit in the CU's 160 KiB.
    static constexpr int oAB = 0;
    static constexpr int oT = oAB + N * LAB;             // T^T: N rows (columns of the pad block go to sTrash)
    static constexpr int oK = oT;                        // [K | d | pad]  M x LK   (inside sT, after S2)
    static constexpr int oT3 = oK + M * LK;              // T3^T           M x N    (inside sT, after S2)
    static_assert(M * LK + M * N <= N * LAB, "sK + sT3 must fit inside sT");
    static constexpr int oP = oT + N * LAB;
    static constexpr int oQ = oP + N * LP;
    static constexpr int oG = oQ + N * LQ;               // M rows
    static constexpr int oTrash = oG + M * LG;           // 2 doubles: where masked-out 16-byte stores go
    static constexpr int total = round_up(oTrash + 2, 2);
    static constexpr bool supported = (N % 2 == 0) && (M % 2 == 0) && (N + M + 1 <= 64) && (total * 8 <= 64 * 1024);
    // masked-out column writes borrow dead buffers: S3's [K|d] columns -> sT3, S6's pad column -> the sT space
    static_assert((M - 1) * LK < M * N && (RB - 1) * LP < N * LAB, "borrowed trash regions too small");
    // per-lane prefetch of [A|B]: 16-byte pairs, round-robin over the wave
    static constexpr int AB_PAIRS = N * NM / 2;
    static constexpr int AB_ROUNDS = (AB_PAIRS + 63) / 64;
};

// acc[r][c] += sum_l X[l*LDX + r] * Y[l*LDY + c]   (X, Y already offset to the block's first row/col)
// Software-pipelined by hand, DEPTH reduction steps deep: the operands of step l+DEPTH-1 are requested before
// the FMAs of step l issue (a ds_read_b128 takes ~130-200 cycles to return with four waves on the CU, one
// step's 16 FMAs only 64), and a scheduling barrier per step keeps the compiler from hoisting every ds_read of
// the fully unrolled loop to the top, which would blow the 256-VGPR budget into AGPR copies.
template <int RBK, int CBK, int L, int LDX, int LDY, int DEPTH = 2>
__device__ __forceinline__ void block_product(const double* __restrict__ X, const double* __restrict__ Y,
                                              double (&acc)[RBK][CBK]) {
    constexpr int D = DEPTH < L ? DEPTH : L;
    v2d xa[D][RBK / 2], yb[D][CBK / 2];
#pragma unroll
    for (int p = 0; p < D - 1; ++p) {
#pragma unroll
        for (int r = 0; r < RBK / 2; ++r) xa[p][r] = *reinterpret_cast<const v2d*>(X + p * LDX + 2 * r);
#pragma unroll
        for (int c = 0; c < CBK / 2; ++c) yb[p][c] = *reinterpret_cast<const v2d*>(Y + p * LDY + 2 * c);
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int cur = l % D, nxt = (l + D - 1) % D;
        if (l + D - 1 < L) {
#pragma unroll
            for (int r = 0; r < RBK / 2; ++r) xa[nxt][r] = *reinterpret_cast<const v2d*>(X + (l + D - 1) * LDX + 2 * r);
#pragma unroll
            for (int c = 0; c < CBK / 2; ++c) yb[nxt][c] = *reinterpret_cast<const v2d*>(Y + (l + D - 1) * LDY + 2 * c);
        }
#pragma unroll
        for (int r = 0; r < RBK; ++r)
#pragma unroll
            for (int c = 0; c < CBK; ++c) {
                const double av = (r & 1) ? xa[cur][r / 2].y : xa[cur][r / 2].x;
                const double bv = (c & 1) ? yb[cur][c / 2].y : yb[cur][c / 2].x;
                acc[r][c] = fma(av, bv, acc[r][c]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Workgroup = kTiledWaves independent wavefronts, each sweeping its own sub-problem out of its own LDS
// slice (no barriers, no sharing).  Packing four of them into one 256-thread workgroup is placement
// control, not cooperation: a workgroup's waves are dealt one per SIMD, whereas 1024 single-wave
// workgroups were observed (in-kernel HW_ID stamps) to land two-on-a-SIMD on ~17 % of the CUs whenever the
// preceding kernel had left the dispatcher's SIMD rotation in an odd state -- and this kernel is
// issue-bound, so two waves on one SIMD run at half speed and the launch takes 1.4x as long.
constexpr int kTiledWaves = 4;

template <int N, int M>
__global__ __launch_bounds__(64 * kTiledWaves) void k_riccati_tiled(int B, int T, const double* __restrict__ tiles,
                                                       const double* __restrict__ mu_arr, double* __restrict__ Kout,
                                                       double* __restrict__ dout, int32_t* __restrict__ singular,
                                                       const int32_t* __restrict__ items,
                                                       const int32_t* __restrict__ n_items, int gains_by_item) {
    using C = TiledCfg<N, M>;
    constexpr int NM = C::NM, RB = C::RB, CB = C::CB, NMP = C::NMP, LAB = C::LAB, LP = C::LP, NP = C::NP;
    constexpr int LQ = C::LQ, LK = C::LK, LG = C::LG;
    const int wave = threadIdx.x >> 6;
    const int slot = blockIdx.x * kTiledWaves + wave;
    if (slot >= (n_items ? *n_items : B)) return;
    const int b = items ? items[slot] : slot;
    if (b >= B) return;
    const int64_t gslot = gains_by_item ? b : slot;   // where K, d of this sub-problem go
    const int lane = threadIdx.x & 63;
    const TileLayout L(N, M);

    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    double* lds = lds_all + wave * C::total;
    double* sAB = lds + C::oAB;
    double* sP = lds + C::oP;
    double* sT = lds + C::oT;
    double* sQ = lds + C::oQ;
    double* sG = lds + C::oG;
    double* sK = lds + C::oK;
    double* sT3 = lds + C::oT3;
    double* sTrash = lds + C::oTrash;

    const double mu = mu_arr[b];
    const double* base = tiles + (int64_t)slot * (T + 1) * L.stride;
    int sing = 0;
    unsigned long long* const stamps = g_stamp_buf;
    unsigned long long t_start = 0;
    if (stamps) t_start = __builtin_amdgcn_s_memrealtime();

    // ---- one-time LDS initialisation: zero everything (pads must stay finite), then P, p from record T
    for (int e = lane; e < C::total; e += 64) lds[e] = 0.0;
    DPILQR_LDS_FENCE();
    {
        const double* rec = base + (int64_t)T * L.stride;
        for (int e = lane; e < N * N; e += 64) {
            const int i = e / N, j = e - i * N;
            sP[i * LP + j] = rec[L.oLxx + e];
        }
        for (int i = lane; i < N; i += 64) sP[i * LP + N] = rec[L.oLx + i];
    }

    // ---- per-lane block coordinates and addresses, constant over the horizon.  The time loop below is
    // branch-free: lanes beyond a phase's block count recompute the last block (identical values to
    // identical addresses), and elements that must not be stored go to sTrash.
    // S1: rows of [A|B]^T (NMP/RB) x cols of [P|p] (LP/CB)
    constexpr int S1_CB = LP / CB, S1_BLOCKS = (NMP / RB) * S1_CB;
    static_assert(S1_BLOCKS <= 64, "S1 needs one block per lane");
    const int s1 = lane < S1_BLOCKS ? lane : S1_BLOCKS - 1;
    const int s1_i0 = (s1 / S1_CB) * RB, s1_j0 = (s1 % S1_CB) * CB;
    const double* s1_x = sAB + s1_i0;
    const double* s1_y = sP + s1_j0;
    // T^T[j0 + c][i0 + r].  The pad block (j0 == n: the p column and padding) has no rows in sT: its lanes
    // collapse every store of this group onto the 16-byte trash slot (offsets multiplied by s1_keep = 0).
    const int s1_keep = (s1_j0 < N) ? 1 : 0;
    double* s1_t = s1_keep ? sT + s1_j0 * LAB + s1_i0 : sTrash;
    // mu enters only B's products (quirk Q6): B^T (P + mu I) = B^T P + mu B^T, added as a correction to the
    // B rows of the stacked product: T2[a][j] += mu * B[j][a].  s1_mu[c] is mu for (B row block, real column j),
    // else 0; s1_b[c] points at B[j][a0..] (any finite row when unused).
    double s1_mu[CB];
    const double* s1_b[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int j = s1_j0 + c;
        const bool reg = (s1_i0 >= N) && (j < N);
        s1_mu[c] = reg ? mu : 0.0;
        s1_b[c] = sAB + (reg ? j : 0) * LAB + (reg ? s1_i0 : 0);
    }
    double* s1_q[RB];                                                // Q_x / Q_u slots of the p column
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int i = s1_i0 + r;
        s1_q[r] = (s1_j0 != N || i >= NM) ? sTrash : (i < N ? sQ + i * LQ + N : sG + (i - N) * LG + M + N);
    }
    int s1_lsrc[RB / 2];                                             // [l_x ; l_u] pairs (safe offset for padded rows)
#pragma unroll
    for (int r = 0; r < RB / 2; ++r) s1_lsrc[r] = L.oLx + ((s1_i0 + 2 * r < NM) ? s1_i0 + 2 * r : 0);
    // S2: (NMP/RB) x (NMP/CB) blocks of [T1;T2][A|B]; region 0 Q_xx, 1 Q_ux, 2 Q_uu, 3 unused (T1 B)
    constexpr int S2_CB = NMP / CB, S2_BLOCKS = (NMP / RB) * S2_CB;
    static_assert(S2_BLOCKS <= 64, "S2 needs one block per lane");
    const int s2 = lane < S2_BLOCKS ? lane : S2_BLOCKS - 1;
    const int s2_i0 = (s2 / S2_CB) * RB, s2_j0 = (s2 % S2_CB) * CB;
    const int s2_reg = (s2_i0 < N) ? ((s2_j0 < N) ? 0 : 3) : ((s2_j0 < N) ? 1 : 2);
    const double* s2_x = sT + s2_i0;
    const double* s2_y = sAB + s2_j0;
    double* s2_dst[RB][CB / 2];
    int s2_src[RB][CB / 2];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int c = 0; c < CB / 2; ++c) {
            const int i = s2_i0 + r, j = s2_j0 + 2 * c;
            const bool ok = (s2_reg != 3) && (i < NM) && (j < NM);
            double* d;
            int o;
            if (s2_reg == 0) { d = sQ + i * LQ + j; o = L.oLxx + i * N + j; }
            else if (s2_reg == 1) { d = sG + (i - N) * LG + M + j; o = L.oLux + (i - N) * L.ldUG + j; }
            else { d = sG + (i - N) * LG + (j - N); o = L.oLuu + (i - N) * L.ldUG + (j - N); }
            s2_dst[r][c] = ok ? d : sTrash;
            s2_src[r][c] = ok ? o : 0;
        }
    // S4: T3^T (M x N) in 2x2 blocks
    constexpr int S4_CB = N / 2, S4_BLOCKS = (M / 2) * S4_CB;
    static_assert(S4_BLOCKS <= 64, "S4 needs one block per lane");
    const int s4 = lane < S4_BLOCKS ? lane : S4_BLOCKS - 1;
    const int s4_c0 = (s4 / S4_CB) * 2, s4_i0 = (s4 % S4_CB) * 2;
    // S5/S6: N x NP in RB x 2 blocks
    constexpr int S5_CB = NP / 2, S5_BLOCKS = (N / RB) * S5_CB;
    static_assert(S5_BLOCKS <= 64, "S5 needs one block per lane");
    const int s5 = lane < S5_BLOCKS ? lane : S5_BLOCKS - 1;
    const int s5_i0 = (s5 / S5_CB) * RB, s5_j0 = (s5 % S5_CB) * 2;
    // S6 per element: P'[i][j] = wa*V[i][j] + wb*V[j][i]  with (wa,wb) = (1/2,1/2) for j < n, (1,0) for the p column
    double* s6_dst[2];          // &sP[i0][j]; the pad column's writes land in the sT space, which is dead by S6
    const double* s6_vt[2];     // &V[j][i0]  (transposed operand; any finite address when unused)
    double s6_wa[2], s6_wb[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int j = s5_j0 + c;
        s6_dst[c] = (j <= N) ? sP + s5_i0 * LP + j : sT;
        s6_vt[c] = (j < N) ? sQ + j * LQ + s5_i0 : sQ;
        s6_wa[c] = (j < N) ? 0.5 : 1.0;
        s6_wb[c] = (j < N) ? 0.5 : 0.0;
    }
    // S3 epilogue: where this lane's column of -X goes in [K | d], and the coalesced copy-out pattern
    double* s3_k = (lane >= M && lane <= M + N) ? sK + (lane - M) : sT3;   // sT3 is not live during S3
    constexpr int K_PAIRS = M * N / 2, K_ROUNDS = (K_PAIRS + 63) / 64;
    int k_in[K_ROUNDS], k_out[K_ROUNDS];
#pragma unroll
    for (int q = 0; q < K_ROUNDS; ++q) {
        const int e2 = min(lane + 64 * q, K_PAIRS - 1), e = 2 * e2;
        k_out[q] = e;
        k_in[q] = (e / N) * LK + (e % N);
    }
    const int d_idx = min(lane, M - 1);
    // [A|B] prefetch: pair q of this lane -> LDS destination
    double* ab_dst[C::AB_ROUNDS];
    int ab_src[C::AB_ROUNDS];
#pragma unroll
    for (int q = 0; q < C::AB_ROUNDS; ++q) {
        const int e = 2 * (lane + 64 * q);
        const int row = e / NM, col = e - row * NM;
        ab_dst[q] = (e < N * NM) ? sAB + row * LAB + col : sTrash;
        ab_src[q] = (e < N * NM) ? L.oA + e : 0;
    }

    // ---- prefetch registers (one record ahead)
    v2d nAB[C::AB_ROUNDS];
    v2d nL[RB][CB / 2];      // l-values of the S2 block
    v2d nLxu[RB / 2];        // [l_x ; l_u][s1_i0 .. s1_i0+RB)
    // Three prefetch points per step, each a full step ahead of its use and each re-filling registers
    // that were consumed the instruction before (no second register set):
    //   after S0 (AB of record t parked in LDS)        -> [A|B] of record t-1
    //   after S1 (l_x, l_u of record t folded in)      -> [l_x; l_u] of record t-1
    //   after S2 (l_xx/l_ux/l_uu of record t folded in) -> the same block of record t-1
    auto prefetch_ab = [&](int t) {
        const double* rec = base + (int64_t)t * L.stride;
#pragma unroll
        for (int q = 0; q < C::AB_ROUNDS; ++q) nAB[q] = *reinterpret_cast<const v2d*>(rec + ab_src[q]);
    };
    auto prefetch_lxu = [&](int t) {
        const double* rec = base + (int64_t)t * L.stride;
#pragma unroll
        for (int r = 0; r < RB / 2; ++r) nLxu[r] = *reinterpret_cast<const v2d*>(rec + s1_lsrc[r]);
    };
    auto prefetch_l = [&](int t) {
        const double* rec = base + (int64_t)t * L.stride;
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB / 2; ++c) nL[r][c] = *reinterpret_cast<const v2d*>(rec + s2_src[r][c]);
    };
    prefetch_ab(T - 1);
    prefetch_lxu(T - 1);
    prefetch_l(T - 1);
    // Drain the prologue's loads here, once.  Otherwise the loop header merges "prologue order" (which the
    // scheduler is free to permute) with the loop's own issue order, and the only wait that is safe for both
    // is vmcnt(0) on every iteration; with nothing pending on entry the in-loop waits stay counted.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched

#ifdef DPILQR_PHASE_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memtime();
#define PHASE_MARK(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[i] += now_ - ph_t; ph_t = now_; }
#else
#define PHASE_MARK(i)
#endif
    for (int t = T - 1; t >= 0; --t) {
        const int tn = t > 0 ? t - 1 : 0;   // record to prefetch (the last step re-reads record 0: harmless)
        // ---- S0: park [A|B] of record t in LDS
#pragma unroll
        for (int q = 0; q < C::AB_ROUNDS; ++q) *reinterpret_cast<v2d*>(ab_dst[q]) = nAB[q];
        DPILQR_LDS_FENCE();
        prefetch_ab(tn);
        PHASE_MARK(0)

        // ---- S1: [T1;T2 | A^T p;B^T p] = [A|B]^T [P|p]
        {
            double acc[RB][CB];
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < CB; ++c) acc[r][c] = 0.0;
            block_product<RB, CB, N, LAB, LP>(s1_x, s1_y, acc);
#pragma unroll
            for (int c = 0; c < CB; ++c) {   // + mu B^T on the B rows, then T^T[j][i']: the orientation S2 wants
#pragma unroll
                for (int r = 0; r < RB; r += 2) {
                    const v2d bt = *reinterpret_cast<const v2d*>(s1_b[c] + r);
                    acc[r][c] = fma(s1_mu[c], bt.x, acc[r][c]);
                    acc[r + 1][c] = fma(s1_mu[c], bt.y, acc[r + 1][c]);
                }
            }
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int r = 0; r < RB; r += 2)
                    *reinterpret_cast<v2d*>(s1_t + s1_keep * (c * LAB + r)) = v2d{acc[r][c], acc[r + 1][c]};
#pragma unroll
            for (int r = 0; r < RB; ++r) {  // p column: Q_x = l_x + A^T p ; Q_u = l_u + B^T p
                const double lv = (r & 1) ? nLxu[r / 2].y : nLxu[r / 2].x;
                *s1_q[r] = lv + acc[r][0];
            }
        }
        DPILQR_LDS_FENCE();
        prefetch_lxu(tn);
        PHASE_MARK(1)

        // ---- S2: [T1;T2][A|B] -> Q_xx, Q_ux, Q_uu
        {
            double acc[RB][CB];
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < CB; ++c) acc[r][c] = 0.0;
            block_product<RB, CB, N, LAB, LAB>(s2_x, s2_y, acc);
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < CB / 2; ++c)
                    *reinterpret_cast<v2d*>(s2_dst[r][c]) = v2d{nL[r][c].x + acc[r][2 * c], nL[r][c].y + acc[r][2 * c + 1]};
        }
        DPILQR_LDS_FENCE();
        prefetch_l(tn);
        PHASE_MARK(2)

        // ---- S3: [K | d] = -Q_uu^-1 [Q_ux | Q_u] : LU with partial pivoting in registers
        {
            const int col = lane < M + N + 1 ? lane : M + N;
            double v[M], invd[M];
#pragma unroll
            for (int r = 0; r < M; ++r) v[r] = sG[r * LG + col];
#pragma unroll
            for (int kk = 0; kk < M; ++kk) {
                int piv = kk;
                double best = fabs(v[kk]);
#pragma unroll
                for (int r = kk + 1; r < M; ++r) {   // first row of maximal |.| (LAPACK idamax)
                    const double av = fabs(v[r]);
                    piv = (av > best) ? r : piv;
                    best = fmax(best, av);
                }
                piv = __builtin_amdgcn_readlane(piv, kk);  // column kk decides the pivot row
                if (piv != kk) {                           // wave-uniform, and rare: keep it a real branch
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int r = kk + 1; r < M; ++r)
                        if (r == piv) {
                            asm volatile("" ::: "memory");   // keep it a scalar branch, not 4 selects per row
                            const double tv = v[r]; v[r] = v[kk]; v[kk] = tv;
                        }
                }
                const double pv = readlane_f64(v[kk], kk);
                if (pv == 0.0) sing = 1;
                // 1/pivot: hardware reciprocal seed + two Newton steps (<= 1 ulp; LAPACK scales by a reciprocal too)
                double inv = __builtin_amdgcn_rcp(pv);
                inv = fma(fma(-pv, inv, 1.0), inv, inv);
                inv = fma(fma(-pv, inv, 1.0), inv, inv);
                invd[kk] = inv;
#pragma unroll
                for (int r = kk + 1; r < M; ++r) {
                    const double l = readlane_f64(v[r], kk) * inv;
                    v[r] = fma(-l, v[kk], v[r]);
                }
            }
#pragma unroll
            for (int r = M - 1; r >= 0; --r) {
                double s = v[r];
#pragma unroll
                for (int c = r + 1; c < M; ++c) s = fma(-readlane_f64(v[r], c), v[c], s);
                v[r] = s * invd[r];
            }
            // K = -X (lanes M..M+N-1 hold its columns), d = -x (lane M+N): into LDS as [K | d]
#pragma unroll
            for (int a = 0; a < M; ++a) s3_k[a * LK] = -v[a];   // lanes outside the range scribble on the dead sT3
        }
        DPILQR_LDS_FENCE();
        // stream K[t] (M x N, contiguous) and d[t] out with unconditional, coalesced stores: every lane
        // stores (surplus lanes repeat the last element) so that the number of outstanding stores is
        // static and the waits on the prefetched loads stay counted instead of collapsing to vmcnt(0)
        {
            double* Kt = Kout + (gslot * T + t) * M * N;
            double* dt_ = dout + (gslot * T + t) * M;
#pragma unroll
            for (int q = 0; q < K_ROUNDS; ++q) store_v2d_nt(Kt + k_out[q], *reinterpret_cast<const v2d*>(sK + k_in[q]));
            store_f64_nt(dt_ + d_idx, sK[d_idx * LK + N]);
        }
        DPILQR_LDS_FENCE();
        PHASE_MARK(3)

        // ---- S4: T3^T[c][i] = sum_a Q_uu[a][c] K[a][i]
        {
            double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
            block_product<2, 2, M, LG, LK>(sG + s4_c0, sK + s4_i0, acc);
            *reinterpret_cast<v2d*>(sT3 + s4_c0 * N + s4_i0) = v2d{acc[0][0], acc[0][1]};
            *reinterpret_cast<v2d*>(sT3 + (s4_c0 + 1) * N + s4_i0) = v2d{acc[1][0], acc[1][1]};
        }
        DPILQR_LDS_FENCE();
        PHASE_MARK(4)

        // ---- S5: V = ((Q_xx + T3 K) + K^T Q_ux) + Q_ux^T K, with [K|d] and [Q_ux|Q_u] carrying the p update
        double vb[RB][2];
        {
            double a1[RB][2], a2[RB][2], a3[RB][2];
#pragma unroll
            for (int r = 0; r < RB; ++r) { a1[r][0] = a1[r][1] = a2[r][0] = a2[r][1] = a3[r][0] = a3[r][1] = 0.0; }
            block_product<RB, 2, M, N, LK>(sT3 + s5_i0, sK + s5_j0, a1);
            block_product<RB, 2, M, LK, LG>(sK + s5_i0, sG + M + s5_j0, a2);
            block_product<RB, 2, M, LG, LK>(sG + M + s5_i0, sK + s5_j0, a3);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const v2d q = *reinterpret_cast<const v2d*>(sQ + (s5_i0 + r) * LQ + s5_j0);
                vb[r][0] = ((q.x + a1[r][0]) + a2[r][0]) + a3[r][0];
                vb[r][1] = ((q.y + a1[r][1]) + a2[r][1]) + a3[r][1];
            }
            DPILQR_LDS_FENCE();
#pragma unroll
            for (int r = 0; r < RB; ++r)
                *reinterpret_cast<v2d*>(sQ + (s5_i0 + r) * LQ + s5_j0) = v2d{vb[r][0], vb[r][1]};
        }
        DPILQR_LDS_FENCE();
        PHASE_MARK(5)

        // ---- S6: P <- (V + V^T)/2 ; p <- V[:, n]
        {
            double pn[RB][2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < RB; ++r) pn[r][c] = fma(s6_wb[c], s6_vt[c][r], s6_wa[c] * vb[r][c]);
            DPILQR_LDS_FENCE();
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < RB; ++r) s6_dst[c][r * LP] = pn[r][c];
        }
        DPILQR_LDS_FENCE();
        PHASE_MARK(6)
    }
    if (singular && sing && lane == 0) singular[b] = 1;
    if (stamps && lane == 0) {
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        stamps[4 * slot + 0] = t_start;
        stamps[4 * slot + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[4 * slot + 2] = hw_id;
        stamps[4 * slot + 3] = xcc_id;
#ifdef DPILQR_PHASE_STAMPS
        for (int i = 0; i < 7; ++i) stamps[4 * B + 8 * slot + i] = ph[i];
#endif
    }
}

}  // namespace dpilqr
