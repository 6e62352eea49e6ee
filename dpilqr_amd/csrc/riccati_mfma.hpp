// riccati_mfma.hpp -- K2 for n_x <= 20: the Riccati backward sweep with one wavefront per sub-problem, up to three
// per SIMD (ilqrSolver._backward_pass, control.py:116-148).  The dominant kernel of the cfg2 benchmark.
//
// Data flow, LDS residency and record prefetching are those of riccati_tiled.hpp (the previous, all-vector-pipe
// version, kept as a fallback).  What differs:
//
//  * the dense products are v_mfma_f64_16x16x4_f64 tiles.  Measured on MI355X (scripts/ubench/mfma_f64.hip): the
//    fp64 MFMA delivers the SAME peak FMA rate as the vector ALU (64 cycles per instruction per SIMD = 16 FMA/clk)
//    and shares that pipe, so it buys issue slots and operand reads (one MFMA replaces 16 vector FMAs and their LDS
//    reads), not arithmetic time.  Every product has the form  C[i][j] = sum_l X[l][i] * Y[l][j]  with X, Y row-major
//    in LDS (rows = reduction index), which is the MFMA operand order:
//        lane (g = lane/16, c = lane%16) supplies A = X[l0+g][i0+c], B = Y[l0+g][j0+c]  (4 reduction rows per MFMA)
//        and owns D[i0 + g + 4v][j0 + c], v = 0..3   (layout verified by the micro-benchmark).
//    So every LDS address of a D element is (lane term) + (compile-time tile/v term); what a lane may store is decided
//    by a few lane predicates and compile-time row ranges.  Reduction lengths that are not multiples of 4 (n_u = 10
//    -> 12) read zero rows kept at the end of [K|d]; reads past a row's logical width wrap into finite neighbouring
//    data and only feed outputs that are dropped.
//  * NS > 0 (the library's own tiles): [A|B] is block diagonal, S1 and S2 need NS terms per output and run on the
//    vector pipe (see the comment at the lane terms below); NS = 0 (plugin tiles): they are MFMA products too.
//  * the pivoted LU keeps all 64 lanes busy, takes its multipliers by DPP row broadcast and skips the pivot search
//    for column-dominant Q_uu (see S3 and lu_eliminate).
//  * a workgroup owns a CU and holds WAVES = 4, 8 or 12 wavefronts (13.5 KB of LDS each), i.e. one, two or three per
//    SIMD, placed deterministically; one wavefront alone gets only half of a SIMD's issue rate.  The live items of a
//    launch are dealt to the CUs in layers (see the kernel's first lines).
//
//   S1  [T1;T2 | A^T p;B^T p] = [A|B]^T [P|p]  (+ mu B^T on the B rows)   dense: 2x2 tiles x 5 k-steps at cfg2
//   S2  [T1;T2] [A|B] + l-values -> Q_xx, Q_ux, Q_uu                        dense: 2x2 x 5
//   S3  LU solve in registers (vector pipe)
//   S4  T3^T = Q_uu-contracted K                                            1x2 x 3
//   S5  a1 = T3 [K|d] ; a2 = [K|d]^T [Q_ux|Q_u] (its transpose supplies Q_ux^T K and Q_ux^T d)   2 x (2x2 x 3)
//   S6  P <- (V + V^T)/2
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "riccati_tiled.hpp"

namespace dpilqr {

typedef double v4d __attribute__((ext_vector_type(4)));

// FUSED: 0 tile records; 1 DoubleInt4D, one Q / R / Q_f; 2 four-state family, per-agent weights.
// HELP (fused forms, launches of at most one wavefront per SIMD): a second wavefront per item evaluates the plugins a step ahead
// into the other of two buffers (see the kernel).
template <int N, int M, int FUSED = 0, bool HELP = false>
struct MfmaCfg {
    static constexpr int NM = N + M;
    static constexpr int NP = N + 1;                   // columns of [P|p], [K|d], [Q_ux|Q_u]
    static constexpr int MK = round_up(M, 4);          // reduction length over controls, zero padded
    static constexpr int LAB = round_up(NM, 2);        // [A|B] rows
    static constexpr int LT = LAB;                     // T^T rows (index i' < NM)
    static constexpr int LP = round_up(NP, 2);
    static constexpr int LQ = LP;                      // [Q_xx | Q_x]
    static constexpr int LG = round_up(M + NP, 2);     // [Q_uu | Q_ux | Q_u]
    static constexpr int LK = LP;                      // [K | d]
    static constexpr int LM = N;                       // a2 = [K|d]^T [Q_ux|Q_u]: rows <= n, columns < n kept
    static constexpr int KROWS = MK + 2;               // [K|d] rows incl. zero rows (a 16-wide tile read may wrap a row)
    static constexpr int T_NM = (NM + 15) / 16, T_NP = (NP + 15) / 16, T_N = (N + 15) / 16, T_M = (M + 15) / 16;
    static constexpr int LTB = LP;                     // block-diagonal variant: T row-major, [i'][j], j <= n
    // LDS carve (doubles): three regions that change owner along the step -- 13.5 KB per wavefront at cfg2, so that
    // three wavefronts fit a SIMD's share of the CU's 160 KB.  Every hand-over happens inside ONE wavefront, whose LDS
    // operations execute in order, so the only rule is "a phase issues all its operand reads before its epilogue
    // stores":
    //   R1: [A|B] (S0 .. S2 operands)  ->  [Q_xx | Q_x] (S2 epilogue .. S5)
    //   R2: [P|p] (S6 .. S1 operand) -> T (S1 epilogue .. S2 operand) -> [K|d] + T3^T (S3 .. S5 operands)
    //       -> V (S5 epilogue, read transposed by S6) -> [P|p]
    //   G : [Q_uu | Q_ux | Q_u] (S1/S2 epilogues .. S5 operands) + Q_x staging (S1 -> S2 epilogue) -> a2 (S5 epilogue,
    //       read transposed).  Its reduction-padding rows >= M therefore hold finite left-overs instead of zeros;
    //       they only ever multiply the zero padding rows of [K|d].
    // FUSED (riccati_mfma_lane.inc, "fused variant"): no [A|B] in LDS; R1 = [Q_xx | Q_x] followed by the step's plugin data:
    // pair gradients [NPAIR][2], pair Hessians [NPAIR][4], their per-agent sums [KA][4], Q + Q^T [16], R + R^T [4], x_f [N]
    static constexpr int F_KA = N / 4, F_NP = F_KA * (F_KA - 1) / 2;
    // and the step's [l_x | l_u] [N + M]
    // (FUSED == 2: Q + Q^T, R + R^T of every agent, and the step's state-dependent entries of every agent's A block [F_KA][4])
    static constexpr int F_W = FUSED == 2 ? F_KA : 1;
    static constexpr int oFG = N * LQ, oFH = oFG + 2 * F_NP, oFD = oFH + 4 * F_NP, oFQQ = oFD + 4 * F_KA, oFRR = oFQQ + 16 * F_W,
                         oFXf = oFRR + 4 * F_W, oFL = oFXf + N, oFA = oFL + round_up(NM, 2),
                         szF1 = round_up(oFA + (FUSED == 2 ? 4 * F_KA : 0), 2);
    // HELP: the step-dependent part (pair derivatives, their sums, [l_x | l_u], the A entries, and the "some pair is near" flag
    // at oFflag) exists twice, FB doubles apart: the helper wavefront fills one copy while the sweep reads the other.  (The copy
    // spans the constants Q + Q^T, R + R^T, x_f too, which are only ever read at their first location.)
    static constexpr int oFflag = szF1;
    static constexpr int FB = HELP ? round_up(szF1 + 2 - oFG, 2) : 0;
    static constexpr int szF = HELP ? round_up(oFG + 2 * FB, 2) : szF1;
    static constexpr int szR1 = FUSED ? szF : round_up(N * LAB > N * LQ ? N * LAB : N * LQ, 2);
    static constexpr int szT0 = N * LT > KROWS * LK + MK * N ? N * LT : KROWS * LK + MK * N;
    static constexpr int szT1 = szT0 > NM * LTB ? szT0 : NM * LTB;
    static constexpr int szR2 = round_up(szT1 > N * LP ? szT1 : N * LP, 2);
    static constexpr int szG0 = MK * LG + round_up(N, 2);
    static constexpr int szG = round_up(szG0 > NP * LM ? szG0 : NP * LM, 2);
    static constexpr int oAB = 0;
    static constexpr int oQ = 0;
    static constexpr int oT = oAB + szR1;
    static constexpr int oK = oT;
    static constexpr int oT3 = oK + KROWS * LK;
    static constexpr int oP = oT;
    static constexpr int oG = oT + szR2;
    static constexpr int oQx = oG + MK * LG;
    static constexpr int oEnd0 = oG + szG;
    // HELP with two row tiles: a2 in a region of its own (riccati_mfma_team_tail.inc)
    static constexpr bool TEAM_TAIL = HELP && (NP > 16) && (NP <= 32) && (M <= 16);
    static constexpr int oA2 = oEnd0;
    static constexpr int oEnd = oEnd0 + (TEAM_TAIL ? round_up(NP * LM, 2) : 0);
    static constexpr int total = round_up(oEnd + 8, 2);    // + store target of idle lanes
    static constexpr bool supported = (N % 4 == 0) && (M % 2 == 0) && (N + M + 1 <= 64) && (total * 8 <= 40 * 1024);
    static constexpr int AB_PAIRS = N * NM / 2;
    static constexpr int AB_ROUNDS = (AB_PAIRS + 63) / 64;
};

// PNS > 0 (riccati_mfma_sweep, "in-sweep production"): what the padded sweep adds per step -- L_xx, L_uu, [l_x | l_u] in the
// padded sizes -- and the producer's scratch, behind the sweep's own LDS (MfmaCfg::total).  PNS = the agents' state dimension.
template <int N, int M, int PNS>
struct InprodCfg {
    static constexpr int PNC = PNS == 6 ? 3 : (PNS == 12 ? 4 : 2);
    static constexpr int KA = PNS > 0 ? ((N / (PNS > 0 ? PNS : 1)) < (M / PNC) ? (N / (PNS > 0 ? PNS : 1)) : (M / PNC)) : 0;   // most agents (N, M) holds
    static constexpr int NPR = KA * (KA - 1) / 2, NPR1 = NPR > 0 ? NPR : 1;
    static constexpr int oLxx = 0;                                  // L_xx [N][N]: zero off the agents' blocks and the coupling blocks
    static constexpr int oLuu = oLxx + N * N;                       // L_uu [M][M]: w_ref (R + R^T) blocks, 1 on the padded diagonal
    static constexpr int oLxu = oLuu + M * M;                       // [l_x | l_u] [N + M]
    static constexpr int oX = oLxu + round_up(N + M, 2);            // x     [n]
    static constexpr int oE = oX + round_up(N, 2);                  // x - x_f
    static constexpr int oU = oE + round_up(N, 2);                  // u     [m]
    static constexpr int oXf = oU + round_up(M, 2);                 // x_f
    static constexpr int oGp = oXf + round_up(N, 2);                // pair gradients [pairs][3]
    static constexpr int oHp = oGp + round_up(3 * NPR1, 2);         // pair Hessians  [pairs][9]
    static constexpr int oQQ = oHp + round_up(9 * NPR1, 2);         // Q + Q^T [agent][PNS * PNS]
    static constexpr int oRR = oQQ + round_up(KA * PNS * PNS, 2);   // R + R^T [agent][PNC * PNC]
    static constexpr int oPair = round_up(oRR + KA * PNC * PNC, 2);    // the pair table (ints)
    static constexpr int total = PNS > 0 ? round_up(oPair + (NPR1 + 1) / 2, 2) : 0;
};

// fp64 vector operations whose FIRST operand is taken from lane L of the executing lane's 16-lane row (DPP
// row_newbcast, the one DPP control the 64-bit ALU has; v_fmac_f64 and v_mov_b64 are the fp64 opcodes that take it
// on gfx950): the LU solve's multipliers and U
// entries reach every lane of the row inside the arithmetic instruction instead of through two v_readlane each.
// The leading s_nop covers "VALU write of a VGPR, then a DPP read of it" (2 wait states), which the compiler's
// hazard recogniser cannot see inside inline assembly.
template <int L>
__device__ __forceinline__ double dpp_fmac_row(double acc, double a, double b) {   // acc + a[lane L of the row] * b, fused
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(L));
    return acc;
}
// ... without the s_nop: for call sites where the DPP operand `a` was written at least two instructions earlier (the caller
// arranges that; an s_nop per multiply-add was a fifth of the LU's instructions)
template <int L>
__device__ __forceinline__ double dpp_fmac_row_nn(double acc, double a, double b) {
    // volatile: these keep their program order among themselves and against the callers' pins, which is what places the
    // multiplies two instructions ahead of the reads
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(L));
    return acc;
}
template <int L>
__device__ __forceinline__ double dpp_mov_row(double a) {   // a[lane L of the row]
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(a), "n"(L));
    return r;
}
#define DPILQR_ROW_SWITCH(FN, ...)                                                                                  \
    switch (l) {                                                                                                    \
    case 0: return FN<0>(__VA_ARGS__); case 1: return FN<1>(__VA_ARGS__); case 2: return FN<2>(__VA_ARGS__);        \
    case 3: return FN<3>(__VA_ARGS__); case 4: return FN<4>(__VA_ARGS__); case 5: return FN<5>(__VA_ARGS__);        \
    case 6: return FN<6>(__VA_ARGS__); case 7: return FN<7>(__VA_ARGS__); case 8: return FN<8>(__VA_ARGS__);        \
    case 9: return FN<9>(__VA_ARGS__); case 10: return FN<10>(__VA_ARGS__); case 11: return FN<11>(__VA_ARGS__);    \
    case 12: return FN<12>(__VA_ARGS__); case 13: return FN<13>(__VA_ARGS__); case 14: return FN<14>(__VA_ARGS__);  \
    default: return FN<15>(__VA_ARGS__);                                                                            \
    }
// `l` is a compile-time constant after unrolling at every call site: the switch folds to one instruction
__device__ __forceinline__ double fmac_row(double acc, double a, double b, int l) { DPILQR_ROW_SWITCH(dpp_fmac_row, acc, a, b) }
__device__ __forceinline__ double fmac_row_nn(double acc, double a, double b, int l) { DPILQR_ROW_SWITCH(dpp_fmac_row_nn, acc, a, b) }
__device__ __forceinline__ double mov_row(double a, int l) { DPILQR_ROW_SWITCH(dpp_mov_row, a) }

__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// acc[it][jt] += sum_{l < KLEN} X[l][16 it + c] * Y[l][16 jt + c']  for TI x TJ output tiles.
// px / py point at X[g][c] / Y[g][c] of this lane (g = lane / 16, c = lane % 16).
template <int TI, int TJ, int KLEN, int LDX, int LDY>
__device__ __forceinline__ void mfma_product(const double* __restrict__ px, const double* __restrict__ py,
                                             v4d (&acc)[TI][TJ]) {
    static_assert(KLEN % 4 == 0, "reduction length must be padded to a multiple of 4");
#pragma unroll
    for (int ks = 0; ks < KLEN / 4; ++ks) {
        double a[TI], b[TJ];
#pragma unroll
        for (int it = 0; it < TI; ++it) a[it] = px[ks * 4 * LDX + 16 * it];
#pragma unroll
        for (int jt = 0; jt < TJ; ++jt) b[jt] = py[ks * 4 * LDY + 16 * jt];
#pragma unroll
        for (int it = 0; it < TI; ++it)
#pragma unroll
            for (int jt = 0; jt < TJ; ++jt) acc[it][jt] = mfma_f64(a[it], b[jt], acc[it][jt]);
    }
}

template <int TI, int TJ>
__device__ __forceinline__ void zero_tiles(v4d (&acc)[TI][TJ]) {
#pragma unroll
    for (int it = 0; it < TI; ++it)
#pragma unroll
        for (int jt = 0; jt < TJ; ++jt) acc[it][jt] = v4d{0.0, 0.0, 0.0, 0.0};
}

// Visit the D elements of one tile (first row `row0`, a compile-time constant after unrolling) whose row
// i = row0 + g + 4v lies in [lo, hi).  f(v, i - g - lo) gets the sub-row index and the lane-independent part of
// (i - lo), so the caller's address is  lane_pointer[(i - g - lo) * LD]  with lane_pointer = buf + g*LD + column.
// A 4-row group entirely inside / outside the range costs no lane test; only a straddling group compares g.
template <typename F>
__device__ __forceinline__ void for_rows(int row0, int lo, int hi, int g, bool col_ok, F&& f) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int r = row0 + 4 * v;           // rows r .. r+3 over g = 0..3
        if (r + 3 < lo || r >= hi) continue;
        const bool inside = (r >= lo) && (r + 3 < hi);
        if (col_ok && (inside || (r + g >= lo && r + g < hi))) f(v, r - lo);
    }
}

// The same walk for epilogues that LOAD: f(v, r - lo, ok) runs for every row group that is not wholly outside, and says
// whether this lane's element is inside.  The caller loads from a clamped address whatever `ok` is and uses the predicate for
// the store only -- a load inside `if (ok)` is an exec-mask region with its own LDS wait, one round trip per element.
template <typename F>
__device__ __forceinline__ void for_rows_p(int row0, int lo, int hi, int g, bool col_ok, F&& f) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int r = row0 + 4 * v;
        if (r + 3 < lo || r >= hi) continue;
        const bool inside = (r >= lo) && (r + 3 < hi);
        f(v, r - lo, col_ok && (inside || (r + g >= lo && r + g < hi)));
    }
}

// In-register LU of S3: lane = column (of Q_uu or of a right-hand side), v[r] = its entry in row r.  SEARCH: with
// dgetf2's partial pivoting; without, for matrices that are known not to need a row swap (see S3).
// KEEPINV = false: the reciprocal pivots are not kept (invd has one element and is not written): the caller's substitution
// recomputes them from U's diagonal with the same instructions (same bits), for 2 m registers less.
// BATCH (ROWLU): the multipliers of three rows ahead of their three updates, no s_nop per update (needs three more doubles
// per lane: not in the variant that runs at 168 registers).
template <bool SEARCH, int M, bool ROWLU, bool KEEPINV = true, bool BATCH = false>
__device__ __forceinline__ void lu_eliminate(double (&v_io)[M], double (&invd)[KEEPINV ? M : 1], int& sing, int* swaps = nullptr) {
    double v[M];   // a local copy: the row swap below must stay a chain of register moves, never an indexed access
#pragma unroll
    for (int r = 0; r < M; ++r) v[r] = v_io[r];
#pragma unroll
    for (int kk = 0; kk < M; ++kk) {
        if constexpr (SEARCH) {
            // partial pivoting (dgetf2's idamax): a row swap is needed iff some |v[r]|, r > kk, is strictly
            // larger than |v[kk]| in column kk.  One max per row decides that; the index search and the
            // swap run only then.
            double mx = 0.0;
#pragma unroll
            for (int r = kk + 1; r < M; ++r) mx = fmax(mx, fabs(v[r]));
            const unsigned long long need = __builtin_amdgcn_ballot_w64(mx > fabs(v[kk]));
            if ((need >> kk) & 1ull) {
                if (swaps) *swaps += 1;   // diagnostic builds only
                int piv = kk;
                double best = fabs(v[kk]);
#pragma unroll
                for (int r = kk + 1; r < M; ++r) {
                    const double av = fabs(v[r]);
                    piv = (av > best) ? r : piv;
                    best = fmax(best, av);
                }
                piv = __builtin_amdgcn_readlane(piv, kk);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = kk + 1; r < M; ++r)
                    if (r == piv) {
                        asm volatile("" ::: "memory");
                        const double tv = v[r]; v[r] = v[kk]; v[kk] = tv;
                    }
            }
        }
        const double pv = ROWLU ? mov_row(v[kk], kk) : readlane_f64(v[kk], kk);
        if (SEARCH && pv == 0.0) sing = 1;
        double inv = __builtin_amdgcn_rcp(pv);
        inv = fma(fma(-pv, inv, 1.0), inv, inv);
        inv = fma(fma(-pv, inv, 1.0), inv, inv);
        if constexpr (KEEPINV) invd[kk] = inv;
        if constexpr (ROWLU) {
            // every lane scales its own entries; the one that matters (minus the multiplier, in lane kk of the row)
            // reaches the row inside the fused multiply-add: fma(-l, v[kk], v[r]) with l = v[r]@kk * inv, bit for bit
            const double ninv = -inv;
            if constexpr (!BATCH) {
#pragma unroll
                for (int r = kk + 1; r < M; ++r) v[r] = fmac_row(v[r], v[r] * ninv, v[kk], kk);
            } else {
            // in groups of three rows: the three multipliers first, then the three updates -- every DPP read then sits two
            // instructions behind the multiply that wrote its operand, and the group needs no s_nop (a shorter last group: one)
#pragma unroll
            for (int r0 = kk + 1; r0 < M; r0 += 3) {
                double tm[3];
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (r0 + q < M) { tm[q] = v[r0 + q] * ninv; asm volatile("" : "+v"(tm[q])); }
                if (r0 + 2 >= M) asm volatile("s_nop 1");
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (r0 + q < M) { v[r0 + q] = fmac_row_nn(v[r0 + q], tm[q], v[kk], kk); asm volatile("" : "+v"(v[r0 + q])); }
            }
            }
        } else {
#pragma unroll
            for (int r = kk + 1; r < M; ++r) {
                const double l = readlane_f64(v[r], kk) * inv;
                v[r] = fma(-l, v[kk], v[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < M; ++r) v_io[r] = v[r];
}

// workgroup barrier behind an LDS-only wait (global prefetches stay in flight): the two wavefronts of an item's team meet here
__device__ __forceinline__ void wave_barrier_all() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The lane id, optionally made opaque to the optimiser: see riccati_mfma_lane.inc.
template <bool ON>
__device__ __forceinline__ int phase_lane(int lane) {
    if constexpr (ON) asm volatile("" : "+v"(lane));
    return lane;
}

// WAVES = 4: one wave per SIMD (launches that cannot fill two).  WAVES = 8: a 512-thread workgroup owns the
// whole CU and waves w and w+4 share a SIMD.  A sweep step alternates a vector-pipe half (S1-S3) with a matrix-pipe
// half (S4-S6); issue priorities by phase (below) keep the wavefronts of a SIMD in different halves.
// WAVES = 12 (block-diagonal variant, launches of more than 2048 items): three wavefronts per SIMD in 168 registers
// each and 13.5 KB of LDS per item (12 items = 158 KB of the CU's 160 KB).  The lane terms are recomputed per phase
// group instead of being kept (riccati_mfma_lane.inc) and the S2 l-values are requested at the top of their own step.
// FUSED (NS = 4, NC = 2, every agent a DoubleIntDynamics4D, one Q / R / Q_f for all agents and items, planar proximity
// cost): the sweep evaluates linearize / quadraticize itself -- no tile records are read.  A = I + dt A_c, B = dt B_c are
// compile-time patterns in dt, so the block products S1 / S2 shrink to the few non-trivial terms (the dropped ones are
// multiplications by exact 0 and 1: the results are those of the record-fed sweep bit for bit); the (X, U)-dependent
// l-values are computed per step in the lanes that add them, with the tile producer's own expressions and orders
// (tiles_wave.hpp), from pair derivatives that ten lanes evaluate at the top of the step.  Per pass and item the sweep
// reads 8 ((T + 1) n_x + T n_u) = 12 160 B of trajectory instead of 535 360 B of records ("fused variant").
struct FusedArgs {
    dpilqr_batch_desc D;
    const double* X;
    const double* U;
};

// (The sweep's body, shared by the two kernels below: FUSED 0 tile records, 1 the DoubleInt4D form, 2 the general four-state form.)
// PAD (dense record-fed form only): the records and the gains have the sizes n_rec <= N, m_rec <= M of a cluster that has no
// instantiation of its own (n_x not a multiple of 4: three six-state agents, CarDynamics3D; or tiny: one six-state agent) and
// the sweep pads them WHILE LOADING into a decoupled (N, M) problem: A = 1 on the diagonal of the padded states, B = 0 in
// their rows and in the padded controls' columns, no cost on padded states, unit L_uu on the diagonal of padded controls.
// Then P, p, Q_ux stay exact zeros in the padded rows and columns, Q_uu is [Q_uu 0; 0 I] (partial pivoting never picks a
// padded row: its entries in the real columns are zeros), the padded rows of [K|d] are zeros, and every sum of the real
// block only gains terms that are exact zeros: the real gains are those of an unpadded sweep.  Only the real m_rec x n_rec
// block of K[t] and the first m_rec entries of d[t] are stored, in the real layout.
// HELP (fused forms, WAVES = 4: launches that cannot give a SIMD a second item -- a job's draining tail, a single small batch):
// a TEAM of two wavefronts per item on one SIMD (wavefronts w and w + 4 of a 512-thread workgroup).  A wavefront alone on a
// SIMD issues at half rate, so whatever the second one takes over is nearly free.  Stage 1: the helper evaluates the step's
// plugin data (fused_step_data: pair derivatives, their per-agent sums, [l_x | l_u], the A entries -- 12 % of a lone
// wavefront's step, profiles/r04_phase_stamps.txt) one step AHEAD into the other of two buffers; the two meet at one
// s_barrier per step.  Same expressions, same order: bit-identical gains.
// PNS > 0 (with PAD): IN-SWEEP PRODUCTION, the record-free form for the cluster sizes and models the fused forms above are not
// written for (the six-state family, CarDynamics3D; PNS = the agents' state dimension, any models of that family, any per-agent
// weights, any n_dims).  There are no tile records: at the top of a step the wavefront evaluates MultiDynamicalModel.linearize
// and GameCost.quadraticize of (X[t], U[t]) itself -- the sparse tile producer's expressions and summation orders (tiles.hpp) --
// straight into the padded [A|B] operand and into padded L_xx / L_uu / [l_x | l_u] arrays in LDS, which the S1 / S2 epilogues
// read where the record-fed form reads its prefetch registers.  The gains are those of the record-fed padded sweep bit for bit;
// per pass and item the sweep reads 8 ((T + 1) n_x + T n_u) bytes of trajectory instead of (T + 1) records.
template <int N, int M, int WAVES, int NS, int NC, int FUSED, bool PAD = false, bool HELP = false, int PNS = 0>
__device__ __forceinline__ void riccati_mfma_sweep(
    int B, int T, const double* __restrict__ tiles, const double* __restrict__ mu_arr, double* __restrict__ Kout,
    double* __restrict__ dout, int32_t* __restrict__ singular, const int32_t* __restrict__ items,
    const int32_t* __restrict__ n_items, int gains_by_item, int n_cus, FusedArgs F, int n_rec = N, int m_rec = M) {
    using C = MfmaCfg<N, M, FUSED, HELP>;
    static_assert(!FUSED || (NS == 4 && NC == 2), "the fused variants are written for the four-state family's blocks");
    static_assert(!HELP || (FUSED != 0 && WAVES == 4), "the helper wavefront serves the fused forms at one item per SIMD");
    static_assert(!PAD || (NS == 0 && FUSED == 0), "padding while loading is written for the dense record-fed form");
    static_assert(PNS == 0 || (NS == 0 && FUSED == 0 && !HELP), "in-sweep production builds on the dense form (padded, or of the exact size: n_rec = N, m_rec = M)");
    constexpr bool INP = PNS > 0;
    using IC = InprodCfg<N, M, PNS>;
    constexpr bool FGEN = (FUSED == 2);   // per-agent weights, per-agent model (DoubleIntDynamics4D / UnicycleDynamics4D)
    constexpr int NM = C::NM, NP = C::NP, MK = C::MK, LAB = C::LAB, LT = C::LT, LP = C::LP, LQ = C::LQ, LG = C::LG;
    constexpr int LK = C::LK, LM = C::LM, T_NM = C::T_NM, T_NP = C::T_NP, T_N = C::T_N, T_M = C::T_M;
    // Items are DEALT to workgroups in layers, not blocked.  A workgroup owns a CU (its LDS), so a launch runs in rounds
    // of n_cus workgroups, and the time of a round is set by the wavefronts per SIMD (measured per 50-step sweep:
    // 327 us with two, 467 us with three -- whether or not the other SIMDs of the CU, or other CUs, are busy).  The grid
    // is sized by the host's upper bound on the item count; the live count n is on the device.  Blocked assignment
    // (slot = WAVES * block + wave) fills the first rounds and leaves a last round that costs as much as a full one.
    // Instead: a LAYER is one wavefront on every SIMD of every CU (4 n_cus items); n items need ceil(n / layer) layers,
    // spread as evenly as possible over the fewest rounds that hold them (five layers at WAVES = 12: a round of three
    // and a round of two, 467 + 327 us instead of 2 x 467).  Surplus wavefronts and workgroups exit at once.
    const int wave_all = threadIdx.x >> 6;
    const bool helper = HELP && wave_all >= 4;          // wavefronts 4..7: the helpers of the items of wavefronts 0..3
    const int wave = HELP ? (wave_all & 3) : wave_all;
    const int n = n_items ? *n_items : B;
    constexpr int LPR = WAVES / 4;   // layers per round
    const int per_layer = 4 * n_cus;
    const int layers = (n + per_layer - 1) / per_layer;
    const int rounds = (layers + LPR - 1) / LPR;
    const int round = (int)blockIdx.x / n_cus, cu = (int)blockIdx.x - round * n_cus;
    if (round >= rounds) return;
    const int lo = layers / rounds, extra = layers - lo * rounds;   // rounds < extra run lo + 1 layers
    const int my_layers = lo + (round < extra ? 1 : 0);
    const int first_layer = round * lo + min(round, extra);
    if (wave >= 4 * my_layers) return;
    const int slot = (first_layer + (wave >> 2)) * per_layer + (wave & 3) * n_cus + cu;
    if (slot >= n) return;
    const int b = items ? items[slot] : slot;
    if (b >= B) return;
    const int64_t gslot = gains_by_item ? b : slot;
    const int lane0 = threadIdx.x & 63;
    const TileLayout L(PAD ? n_rec : N, PAD ? m_rec : M);

    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    double* lds = lds_all + wave * (C::total + IC::total);
    double* sAB = lds + C::oAB;
    double* sT = lds + C::oT;
    double* sK = lds + C::oK;          // [K | d], KROWS rows (rows >= M zero), after S2
    double* sT3 = lds + C::oT3;        // T3^T, MK rows, after S2
    double* sP = lds + C::oP;
    double* sQ = lds + C::oQ;
    double* sG = lds + C::oG;
    double* sQx = lds + C::oQx;        // Q_x between the S1 and S2 epilogues
    double* sMt = C::TEAM_TAIL ? lds + C::oA2 : sG;   // a2, after the S5 products

    const double mu = mu_arr[b];
    const double f_radius = FUSED ? F.D.radius[(int64_t)b * F.D.radius_bstride] : 0.0;
    const double* base = (FUSED || INP) ? nullptr : tiles + (int64_t)slot * (T + 1) * L.stride;
    int sing = 0;
    unsigned long long* const stamps = g_stamp_buf;
    unsigned long long t_start = 0;
    if (stamps) t_start = __builtin_amdgcn_s_memrealtime();

    constexpr bool ROWLU = (M < 16) && (4 * (16 - M) >= NP);
    constexpr int K_PAIRS = M * N / 2, K_ROUNDS = (K_PAIRS + 63) / 64;
    constexpr bool BD = NS > 0;
    constexpr int KA = BD ? N / (BD ? NS : 1) : 1, LPA = 64 / KA, NSC = NS + NC;
    constexpr int CPL = 2 * ((NP + 2 * LPA - 1) / (2 * LPA));
    constexpr int RPL = (NM + LPA - 1) / LPA;
    constexpr int LTB = C::LTB;
    static_assert(!BD || (NS % 2 == 0 && NC % 2 == 0 && N % 2 == 0 && M % 2 == 0 && CPL == 2),
                  "block-diagonal variant: 16-byte vector accesses need even block sizes");
    v2d nbL[RPL][NSC / 2 > 0 ? NSC / 2 : 1];
    v2d nbX[NSC / 2 > 0 ? NSC / 2 : 1];
    // WAVES = 12 (three wavefronts per SIMD, 168 registers each): the S2 l-values are requested at the top of their own
    // step instead of a step ahead, so that they are not held across the register-hungry phases S3-S6
    constexpr bool LATE_L = BD && (WAVES == 12);
    v2d nAB[C::AB_ROUNDS];
    double nL[T_NM][T_NM][4];
    double nLxu[T_NM][4];
    // see riccati_mfma_lane.inc.  PAD at N >= 20: the offset codes of every prefetched element (some forty per lane) must not
    // be hoisted out of the horizon loop either: kept, they spilled 108 registers at two wavefronts per SIMD (three quadcopters:
    // 1.63 ms per 2048 items against 0.83 ms).  The smaller sizes have the registers and are faster with the terms kept
    // (one quadcopter 0.13 against 0.21 ms).
    constexpr bool REMAT = (WAVES == 12) || (PAD && N >= 20);
    v2d pf[2];   // FUSED: this lane's share of (X[t], U[t]), one step ahead
    bool f_prox = false;   // FUSED: some pair of the current step is within the radius (wave uniform)
    // FUSED: the part of this lane's S2 l-values that does not depend on (X, U) -- w_ref (Q + Q^T), w_ref (R + R^T) on the
    // agent's own blocks -- and where the part that does (a pair Hessian, or the agent's sum of them) is found
    double lvc[FUSED ? RPL : 1][FUSED ? NSC : 1];
    int lv_h[FUSED ? RPL : 1];          // offset into the step's Hessian data (doubles from sFH), -1: no proximity part
    bool lv_neg[FUSED ? RPL : 1];
    int t_cur = T;      // the step whose plugin data the lane terms point at (HELP: selects the buffer)
    if constexpr (HELP) {
        if (helper) {
            const int lane = lane0;
            wave_barrier_all();       // (A) the sweep's wavefront has zeroed the item's LDS and put the constants in place
            {
                t_cur = T - 1;
#include "riccati_mfma_lane.inc"
                fused_prefetch(T - 1);
            }
            for (int t = T - 1; t >= 0; --t) {
                t_cur = t;
#include "riccati_mfma_lane.inc"
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this step's share of (X[t], U[t])
                fused_step_data(t, nullptr);
                if (lane == 0) sFflag[0] = f_prox ? 1.0 : 0.0;
                if (t > 0) fused_prefetch(t - 1);
                if constexpr (C::TEAM_TAIL) {
                    // ... then the helper takes the second row tile of S4 .. S6 of the step the sweep is in, t + 1
                    if (t < T - 1) {
                        wave_barrier_all();   // [K | d] of step t + 1 is in place (the sweep's S3)
                        constexpr int IT = 1;
#include "riccati_mfma_team_tail.inc"
                    }
                }
                wave_barrier_all();   // step t's data are in place; the sweep has finished step t + 1 (the buffer written next)
            }
            if constexpr (C::TEAM_TAIL) {     // the last step's (t = 0) second row tile
                t_cur = 0;
#include "riccati_mfma_lane.inc"
                wave_barrier_all();
                constexpr int IT = 1;
#include "riccati_mfma_team_tail.inc"
                wave_barrier_all();
            }
            return;
        }
    }
    if constexpr (FUSED) {
        const ItemParams IP = item_params(F.D, b);
        constexpr int FKA_ = C::F_KA;
        const int ag_ = min(lane0 / LPA, KA - 1), sub2_ = min(lane0 - (lane0 / LPA) * LPA, (NM - 1) / RPL);
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
#pragma unroll
            for (int c = 0; c < NSC; ++c) lvc[r][c] = 0.0;
            lv_h[r] = -1; lv_neg[r] = false;
            const int ip = min(RPL * sub2_ + r, NM - 1);
            if (ip < N) {
                const int ar = ip >> 2, li = ip & 3;
                if (ar == ag_) {
                    const double* Qa = IP.Q + (FGEN ? 16 * ag_ : 0);
#pragma unroll
                    for (int c = 0; c < NS; ++c) lvc[r][c] = F.D.w_ref * (Qa[li * 4 + c] + Qa[c * 4 + li]);
                }
                if (FKA_ > 1 && li < 2) {
                    lv_neg[r] = ar != ag_;
                    lv_h[r] = (ar == ag_) ? (C::oFD - C::oFH) + ag_ * 4 + li * 2
                                          : ((ar < ag_) ? pair_index(ar, ag_, FKA_) : pair_index(ag_, ar, FKA_)) * 4 + li * 2;
                }
            } else {
                const int a = ip - N;
                if ((a >> 1) == ag_) {
                    const double* Ra = IP.R + (FGEN ? 4 * ag_ : 0);
                    lvc[r][NS] = F.D.w_ref * (Ra[(a & 1) * 2] + Ra[a & 1]);
                    lvc[r][NS + 1] = F.D.w_ref * (Ra[(a & 1) * 2 + 1] + Ra[2 + (a & 1)]);
                }
            }
        }
    }
#include "riccati_mfma_inprod.inc"
    {
    const int lane = lane0;
    for (int e = lane; e < C::total + IC::total; e += 64) lds[e] = 0.0;
    DPILQR_LDS_FENCE();
    if constexpr (INP) {
        // the constants; then the terminal condition P = l_xx(T), p = l_x(T) (control.py:125-129) from the producer's own arrays
        inp_constants();
        inp_prefetch(T);
        inp_produce(T, true);
        for (int e = lane; e < N * N; e += 64) {
            const int i = e / N, j = e - i * N;
            sP[i * LP + j] = sLxx[e];
        }
        for (int i = lane; i < N; i += 64) sP[i * LP + N] = sLxu[i];
        DPILQR_LDS_FENCE();
        inp_lxx_blocks(false, true);     // the agents' whole blocks again, with Q in place of Q_f
        inp_prefetch(T - 1);
    } else if constexpr (!FUSED) {
        const double* rec = base + (int64_t)T * L.stride;
        for (int e = lane; e < N * N; e += 64) {
            const int i = e / N, j = e - i * N;
            if constexpr (PAD) {
                const double val = rec[L.oLxx + min(i, n_rec - 1) * n_rec + min(j, n_rec - 1)];
                sP[i * LP + j] = (i < n_rec && j < n_rec) ? val : 0.0;
            } else {
            sP[i * LP + j] = rec[L.oLxx + e];
            }
        }
        for (int i = lane; i < N; i += 64) {
            const double val = rec[L.oLx + (PAD ? min(i, n_rec - 1) : i)];
            sP[i * LP + N] = (!PAD || i < n_rec) ? val : 0.0;
        }
    }

#include "riccati_mfma_lane.inc"
    if constexpr (FUSED) {
        // symmetrised weights and the goal, once; then the terminal condition p = l_x(T), P = l_xx(T) (control.py:125-129)
        // with the tile producer's expressions (tiles_wave.hpp phases A2-B2, Q_f in place of Q)
        const ItemParams IP = item_params(F.D, b);
        if constexpr (FGEN) {
            for (int e = lane; e < 16 * FKA; e += 64) {
                const int a = e >> 4, q = e & 15;
                sFQQ[e] = IP.Q[16 * a + q] + IP.Q[16 * a + (q & 3) * 4 + (q >> 2)];
            }
            for (int e = lane; e < 4 * FKA; e += 64) {
                const int a = e >> 2, q = e & 3;
                sFRR[e] = IP.R[4 * a + q] + IP.R[4 * a + (q & 1) * 2 + (q >> 1)];
            }
        } else {
        if (lane < 16) sFQQ[lane] = IP.Q[lane] + IP.Q[(lane & 3) * 4 + (lane >> 2)];
        if (lane < 4) sFRR[lane] = IP.R[lane] + IP.R[(lane & 1) * 2 + (lane >> 1)];
        }
        for (int e = lane; e < N; e += 64) sFXf[e] = IP.xf[e];
        DPILQR_LDS_FENCE();
        if constexpr (HELP) wave_barrier_all();   // (A) the helper may start on step T - 1
        fused_prefetch(T);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        fused_step_data(T, IP.Qf);           // pair derivatives at X[T]; [l_x | .] with Q_f
        for (int e = lane; e < N * N; e += 64) {
            const int i = e / N, j = e - i * N, ai = i >> 2, li = i & 3, aj = j >> 2, lj = j & 3;
            double val = 0.0;
            if (ai == aj) {
                const double* Qfa = IP.Qf + (FGEN ? 16 * ai : 0);
                val = F.D.w_ref * (Qfa[li * 4 + lj] + Qfa[lj * 4 + li]);
            }
            if (FKA > 1 && li < 2 && lj < 2 && f_prox) {
                double acc = 0.0;
                if (ai == aj) acc = sFD[ai * 4 + li * 2 + lj];
                else acc += -sFH[((ai < aj) ? pair_index(ai, aj, FKA) : pair_index(aj, ai, FKA)) * 4 + li * 2 + lj];
                val += F.D.w_prox * acc;
            }
            sP[i * LP + j] = val;
        }
        for (int j = lane; j < N; j += 64) sP[j * LP + N] = sFL[j];
        DPILQR_LDS_FENCE();
        if constexpr (!HELP) fused_prefetch(T - 1);
    } else if constexpr (INP) {
    } else {
    prefetch_ab(T - 1);
    if constexpr (BD) {
        prefetch_bd_x(T - 1);
        if constexpr (!LATE_L) prefetch_bd_l(T - 1);
    } else {
        prefetch_lxu(T - 1);
        prefetch_l(T - 1);
    }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see riccati_tiled.hpp
    }

#ifdef DPILQR_PHASE_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memtime();
#define MPHASE(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[i] += now_ - ph_t; ph_t = now_; }
#elif defined(DPILQR_PHASE_MARKS)   // assembly listings only: where each phase ends (hipcc -S, scripts/isa_census.py)
#define MPHASE(i) asm volatile("; ==== end of phase " #i);
#else
#define MPHASE(i)
#endif
    for (int t = T - 1; t >= 0; --t) {
        const int tn = t > 0 ? t - 1 : 0;
        t_cur = t;
        {   // ---- S0, S1
        const int lane = phase_lane<REMAT>(lane0);
#include "riccati_mfma_lane.inc"
        if constexpr (HELP) {
            // the helper wavefront has put this step's plugin data into buffer t & 1 during the previous step
            wave_barrier_all();
            f_prox = __builtin_amdgcn_readfirstlane((int)(sFflag[0] != 0.0)) != 0;
        } else if constexpr (FUSED) {
            // the step's pair derivatives and [l_x | l_u]; then the next step's share of the trajectory is requested
            fused_step_data(t, nullptr);
            fused_prefetch(tn);
        } else if constexpr (INP) {
            // this step's [A|B], L_xx, [l_x | l_u] from (X[t], U[t]); then the next step's share of the trajectory is requested
            inp_produce(t, false);
            inp_prefetch(tn);
        } else {
#pragma unroll
        for (int q = 0; q < C::AB_ROUNDS; ++q) *reinterpret_cast<v2d*>(ab_dst[q]) = nAB[q];
        DPILQR_LDS_FENCE();
        if constexpr (LATE_L) prefetch_bd_l(t);
        prefetch_ab(tn);
        }
        MPHASE(0)

        // The two wavefronts of a SIMD should be in DIFFERENT halves of a step -- one in the vector-pipe phases (S1-S3:
        // the block products and the LU with its dependent chains), the other in the matrix-pipe phases (S4-S6), whose
        // MFMAs need few issue slots.  A start offset does not achieve that (measured: nothing), issue priority does:
        // the wavefront in S1-S3 wins arbitration and the other's MFMAs fill in behind it.  327 us instead of 342 us
        // per 2048-item launch.
        __builtin_amdgcn_s_setprio(1);
        // ---- S1: [A|B]^T [P|p]
        if constexpr (FUSED) {
            // [A|B]^T [P|p] for A = I + dt A_c, B = dt B_c of the double integrator: rows 0, 1 of A^T P are rows 0, 1 of P;
            // rows 2, 3 are dt P_0 + P_2, dt P_1 + P_3 (product rounded, then the sum: what the four-term chain of the
            // record-fed sweep computes, whose other terms are exact zeros); B^T P = dt [P_2; P_3]
            const double fdt = F.D.dt;
            double acc[NSC][CPL], pr[NS][CPL];
#pragma unroll
            for (int l = 0; l < NS; ++l) {
                const v2d v = *reinterpret_cast<const v2d*>(bP + l * LP);
                pr[l][0] = v.x; pr[l][1] = v.y;
            }
            // FGEN: the agent's A block is I + dt A_c with A_c's entries (0,2), (1,2), (0,3), (1,3) free (Unicycle4D: dt cos, dt sin,
            // -dt v sin, dt v cos; DoubleInt4D: dt, 0, 0, dt), the step's values in sFA.  Row 2 of A^T P is the record-fed sweep's
            // four-term chain a02 P_0, fma(a12, P_1, .), fma(1, P_2, .), fma(0, P_3, .) with the last two written as what they are
            double fa[4] = {0.0, 0.0, 0.0, 0.0};
            if constexpr (FGEN) {
                const v2d a0 = *reinterpret_cast<const v2d*>(sFA + 4 * ag), a1 = *reinterpret_cast<const v2d*>(sFA + 4 * ag + 2);
                fa[0] = a0.x; fa[1] = a0.y; fa[2] = a1.x; fa[3] = a1.y;   // a02, a12, a03, a13
            }
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                acc[0][c] = pr[0][c];
                acc[1][c] = pr[1][c];
                if constexpr (FGEN) {
                    acc[2][c] = fma(fa[1], pr[1][c], fa[0] * pr[0][c]) + pr[2][c];
                    acc[3][c] = fma(fa[3], pr[1][c], fa[2] * pr[0][c]) + pr[3][c];
                } else {
                acc[2][c] = fdt * pr[0][c] + pr[2][c];
                acc[3][c] = fdt * pr[1][c] + pr[3][c];
                }
                acc[4][c] = fdt * pr[2][c];
                acc[5][c] = fdt * pr[3][c];
                // T2 rows: + mu B[j][c] with B[j][2 ag + q] = dt iff j = 4 ag + 2 + q   (quirk Q6)
#pragma unroll
                for (int q = 0; q < NC; ++q) acc[NS + q][c] = fma(mu, (j0 + c == 4 * ag + 2 + q) ? fdt : 0.0, acc[NS + q][c]);
            }
#pragma unroll
            for (int r = 0; r < NS; ++r) *reinterpret_cast<v2d*>(bTa + r * LTB) = v2d{acc[r][0], acc[r][1]};
#pragma unroll
            for (int r = 0; r < NC; ++r) *reinterpret_cast<v2d*>(bTb + r * LTB) = v2d{acc[NS + r][0], acc[NS + r][1]};
            if (bPcol) {   // Q_x = l_x + A^T p ; Q_u = l_u + B^T p
#pragma unroll
                for (int r = 0; r < NS; ++r) sQx[NS * ag + r] = sFL[NS * ag + r] + acc[r][0];
#pragma unroll
                for (int r = 0; r < NC; ++r) sG[(NC * ag + r) * LG + M + N] = sFL[N + NC * ag + r] + acc[NS + r][0];
            }
        } else if constexpr (BD) {
            double acc[NSC][CPL];
#pragma unroll
            for (int l = 0; l < NS; ++l) {
                double ab[NSC], pr[CPL];
#pragma unroll
                for (int q = 0; q < NS / 2; ++q) {
                    const v2d v = *reinterpret_cast<const v2d*>(bABa + l * LAB + 2 * q);
                    ab[2 * q] = v.x; ab[2 * q + 1] = v.y;
                }
#pragma unroll
                for (int q = 0; q < NC / 2; ++q) {
                    const v2d v = *reinterpret_cast<const v2d*>(bABb + l * LAB + 2 * q);
                    ab[NS + 2 * q] = v.x; ab[NS + 2 * q + 1] = v.y;
                }
                {
                    const v2d v = *reinterpret_cast<const v2d*>(bP + l * LP);
                    pr[0] = v.x; pr[1] = v.y;
                }
#pragma unroll
                for (int r = 0; r < NSC; ++r)
#pragma unroll
                    for (int c = 0; c < CPL; ++c) acc[r][c] = (l == 0) ? ab[r] * pr[c] : fma(ab[r], pr[c], acc[r][c]);
            }
            // T2 rows: + mu B[j][c]   (quirk Q6: B^T (P + mu I) = B^T P + mu B^T)
#pragma unroll
            for (int c = 0; c < CPL; ++c)
#pragma unroll
                for (int q = 0; q < NC / 2; ++q) {
                    const v2d v = *reinterpret_cast<const v2d*>(bBmu[c] + 2 * q);
                    acc[NS + 2 * q][c] = fma(mu, bMuOn[c] ? v.x : 0.0, acc[NS + 2 * q][c]);
                    acc[NS + 2 * q + 1][c] = fma(mu, bMuOn[c] ? v.y : 0.0, acc[NS + 2 * q + 1][c]);
                }
#pragma unroll
            for (int r = 0; r < NS; ++r) *reinterpret_cast<v2d*>(bTa + r * LTB) = v2d{acc[r][0], acc[r][1]};
#pragma unroll
            for (int r = 0; r < NC; ++r) *reinterpret_cast<v2d*>(bTb + r * LTB) = v2d{acc[NS + r][0], acc[NS + r][1]};
            // p column: Q_x = l_x + A^T p ; Q_u = l_u + B^T p
            if (bPcol) {
#pragma unroll
                for (int r = 0; r < NS; ++r) sQx[NS * ag + r] = ((r & 1) ? nbX[r / 2].y : nbX[r / 2].x) + acc[r][0];
#pragma unroll
                for (int r = 0; r < NC; ++r)
                    sG[(NC * ag + r) * LG + M + N] = ((r & 1) ? nbX[NS / 2 + r / 2].y : nbX[NS / 2 + r / 2].x) + acc[NS + r][0];
            }
        } else {
            v4d acc[T_NM][T_NP];
            zero_tiles(acc);
            mfma_product<T_NM, T_NP, N, LAB, LP>(pAB, pP, acc);
#pragma unroll
            for (int it = 0; it < T_NM; ++it)
#pragma unroll
                for (int jt = 0; jt < T_NP; ++jt) {
                    // T1 rows: T^T[j][i'] = acc
                    for_rows(16 * it, 0, N, g, colN[jt], [&](int v, int r) { dTt[16 * jt * LT + r] = acc[it][jt][v]; });
                    // T2 rows: + mu B[j][a]   (quirk Q6: B^T (P + mu I) = B^T P + mu B^T)
                    for_rows(16 * it, N, NM, g, colN[jt], [&](int v, int r) {
                        dTt[16 * jt * LT + N + r] = fma(mu, dBt[16 * jt * LAB + N + r], acc[it][jt][v]);
                    });
                    // p column: Q_x = l_x + A^T p ; Q_u = l_u + B^T p
                    if (16 * jt <= N && N < 16 * jt + 16) {
                        for_rows(16 * it, 0, N, g, colP[jt], [&](int v, int r) {
                            sQx[g + r] = (INP ? sLxu[g + r] : nLxu[it][v]) + acc[it][jt][v];
                        });
                        for_rows(16 * it, N, NM, g, colP[jt], [&](int v, int r) {
                            dG[M + N + r * LG] = (INP ? sLxu[N + g + r] : nLxu[it][v]) + acc[it][jt][v];
                        });
                    }
                }
        }
        DPILQR_LDS_FENCE();
        if constexpr (!BD && !FUSED && !INP) prefetch_lxu(tn);
        MPHASE(1)
        }
        {
        const int lane = phase_lane<REMAT>(lane0);
#include "riccati_mfma_lane.inc"

        // ---- S2: [T1;T2][A|B] -> Q_xx (rows < n, cols < n), [Q_uu | Q_ux] (rows >= n); the T1 B block is dropped
        if constexpr (FUSED) {
            // [T1;T2] [A|B] with the same patterns: columns 0, 1 of T A are T's, columns 2, 3 are dt T_0 + T_2, dt T_1 + T_3,
            // T B = dt [T_2 T_3]; then the step's l-values, formed here instead of being read
            const double fdt = F.D.dt, wp = F.D.w_prox;
            double fa2[4] = {0.0, 0.0, 0.0, 0.0};
            if constexpr (FGEN) {
                const v2d a0 = *reinterpret_cast<const v2d*>(sFA + 4 * ag), a1 = *reinterpret_cast<const v2d*>(sFA + 4 * ag + 2);
                fa2[0] = a0.x; fa2[1] = a0.y; fa2[2] = a1.x; fa2[3] = a1.y;
            }
            // (the Hessian entries of all rows first, from a clamped offset: no load behind a per-lane test)
            v2d hh[RPL];
            const bool with_h = FKA > 1 && f_prox;
            if (with_h) {
#pragma unroll
                for (int r = 0; r < RPL; ++r) hh[r] = *reinterpret_cast<const v2d*>(sFH + max(lv_h[r], 0));
            }
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                double tv[NS], acc[NSC], lv[NSC];
#pragma unroll
                for (int q = 0; q < NS / 2; ++q) {
                    const v2d v = *reinterpret_cast<const v2d*>(bT2[r] + 2 * q);
                    tv[2 * q] = v.x; tv[2 * q + 1] = v.y;
                }
                acc[0] = tv[0]; acc[1] = tv[1];
                if constexpr (FGEN) {   // columns 2, 3 of T A: the chain over A's rows 0, 1, then the unit entry (see S1)
                    acc[2] = fma(tv[1], fa2[1], tv[0] * fa2[0]) + tv[2];
                    acc[3] = fma(tv[1], fa2[3], tv[0] * fa2[2]) + tv[3];
                } else {
                acc[2] = tv[0] * fdt + tv[2]; acc[3] = tv[1] * fdt + tv[3];
                }
                acc[4] = tv[2] * fdt; acc[5] = tv[3] * fdt;
#pragma unroll
                for (int c = 0; c < NSC; ++c) lv[c] = lvc[r][c];
                if (with_h) {     // the pair Hessian (or the agent's sum of them) on the position entries
                    const bool on = lv_h[r] >= 0;
                    const double l0 = lv[0] + wp * (lv_neg[r] ? -hh[r].x : hh[r].x), l1 = lv[1] + wp * (lv_neg[r] ? -hh[r].y : hh[r].y);
                    lv[0] = on ? l0 : lv[0];
                    lv[1] = on ? l1 : lv[1];
                }
#pragma unroll
                for (int q = 0; q < NS / 2; ++q)
                    *reinterpret_cast<v2d*>(bDa[r] + 2 * q) = v2d{lv[2 * q] + acc[2 * q], lv[2 * q + 1] + acc[2 * q + 1]};
                *reinterpret_cast<v2d*>(bDb[r]) = v2d{lv[NS] + acc[NS], lv[NS + 1] + acc[NS + 1]};
            }
        } else if constexpr (BD) {
            double acc[RPL][NSC];
            double tv[RPL][NS];
            {
#pragma unroll
                for (int r = 0; r < RPL; ++r)
#pragma unroll
                    for (int q = 0; q < NS / 2; ++q) {
                        const v2d v = *reinterpret_cast<const v2d*>(bT2[r] + 2 * q);
                        tv[r][2 * q] = v.x; tv[r][2 * q + 1] = v.y;
                    }
            }
#pragma unroll
            for (int l = 0; l < NS; ++l) {
                double ab[NSC];
#pragma unroll
                for (int q = 0; q < NS / 2; ++q) {
                    const v2d v = *reinterpret_cast<const v2d*>(bABa + l * LAB + 2 * q);
                    ab[2 * q] = v.x; ab[2 * q + 1] = v.y;
                }
#pragma unroll
                for (int q = 0; q < NC / 2; ++q) {
                    const v2d v = *reinterpret_cast<const v2d*>(bABb + l * LAB + 2 * q);
                    ab[NS + 2 * q] = v.x; ab[NS + 2 * q + 1] = v.y;
                }
#pragma unroll
                for (int r = 0; r < RPL; ++r) {
                    const double tvl = tv[r][l];
#pragma unroll
                    for (int c = 0; c < NSC; ++c) acc[r][c] = (l == 0) ? tvl * ab[c] : fma(tvl, ab[c], acc[r][c]);
                }
            }
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
#pragma unroll
                for (int q = 0; q < NS / 2; ++q)
                    *reinterpret_cast<v2d*>(bDa[r] + 2 * q) = v2d{nbL[r][q].x + acc[r][2 * q], nbL[r][q].y + acc[r][2 * q + 1]};
#pragma unroll
                for (int q = 0; q < NC / 2; ++q)
                    *reinterpret_cast<v2d*>(bDb[r] + 2 * q) =
                        v2d{nbL[r][NS / 2 + q].x + acc[r][NS + 2 * q], nbL[r][NS / 2 + q].y + acc[r][NS + 2 * q + 1]};
            }
        } else {
            v4d acc[T_NM][T_NM];
            zero_tiles(acc);
            mfma_product<T_NM, T_NM, N, LT, LAB>(pT, pAB, acc);
#pragma unroll
            for (int it = 0; it < T_NM; ++it)
#pragma unroll
                for (int jt = 0; jt < T_NM; ++jt) {
                    if constexpr (INP) {   // the l-values from the producer's arrays: L_xx[i][j]; row a of [L_ux | L_uu] (L_ux = 0)
                        const int j = 16 * jt + c16;
                        for_rows(16 * it, 0, N, g, colN[jt], [&](int v, int r) {
                            dQ[16 * jt + r * LQ] = sLxx[(g + r) * N + j] + acc[it][jt][v];
                        });
                        for_rows(16 * it, N, NM, g, colNM[jt], [&](int v, int r) {
                            const double luu = sLuu[(g + r) * M + max(j - N, 0)];
                            dG[colG[jt] + r * LG] = (j >= N ? luu : 0.0) + acc[it][jt][v];
                        });
                    } else {
                    for_rows(16 * it, 0, N, g, colN[jt], [&](int v, int r) { dQ[16 * jt + r * LQ] = nL[it][jt][v] + acc[it][jt][v]; });
                    for_rows(16 * it, N, NM, g, colNM[jt], [&](int v, int r) { dG[colG[jt] + r * LG] = nL[it][jt][v] + acc[it][jt][v]; });
                    }
                }
        }
        DPILQR_LDS_FENCE();
        if (lane < N) sQ[lane * LQ + N] = sQx[lane];   // Q_x joins Q_xx now that [A|B] is dead
        if constexpr (FUSED || INP) {} else if constexpr (BD) { prefetch_bd_x(tn); if constexpr (!LATE_L) prefetch_bd_l(tn); } else prefetch_l(tn);
        // sT is dead from here on and becomes [K | d] + T3^T: the reduction-padding rows of [K | d] must read as zero
        for (int e = lane; e < (C::KROWS - M) * LK; e += 64) sK[M * LK + e] = 0.0;
        MPHASE(2)
        }
        {
        const int lane = phase_lane<REMAT>(lane0);
#include "riccati_mfma_lane.inc"

        // ---- S3: [K | d] = -Q_uu^-1 [Q_ux | Q_u] : LU with partial pivoting in registers (vector pipe)
        {
            const int col = s3_col;
            double v[M], invd[M];
#pragma unroll
            for (int r = 0; r < M; ++r) v[r] = sG[r * LG + col];
            // Shortcut for the common case.  If every column of Q_uu is strictly diagonally dominant, partial pivoting
            // (dgetf2's idamax) moves no row: the diagonal is the strict maximum of its column, and the dominance gap
            // |a_jj| - sum_{i != j} |a_ij| of a column does not shrink from one Schur complement to the next.  The
            // 2^-20 relative margin dwarfs the rounding of ten eliminations (element growth <= 2), so the test decides
            // exactly what the per-column search would decide, and the elimination then runs without the search -- the
            // ten serial max chains are what the LU's critical path could least afford.  NaNs fail the test.
            double colsum = 0.0;
#pragma unroll
            for (int r = 0; r < M; ++r) colsum = colsum + fabs(v[r]);
            const double dg = fabs(sG[s3_dg]);
            const bool dominant = !s3_lhs || dg * (1.0 - 0x1p-20) > colsum - dg;
            const bool no_swaps = __builtin_amdgcn_ballot_w64(!dominant) == 0ull;
            constexpr bool LUB = ROWLU && (WAVES == 4);   // one wavefront per SIMD: the s_nops cost their full issue time there (and the two- and three-per-SIMD variants have no registers to spare)
            if (no_swaps) lu_eliminate<false, M, ROWLU, true, LUB>(v, invd, sing); else lu_eliminate<true, M, ROWLU, true, LUB>(v, invd, sing);
            if constexpr (ROWLU) {
                double nv[M];   // minus the solved rows: fma(-U, x, s) == fma(U, -x, s)
#pragma unroll
                for (int r = M - 1; r >= 0; --r) {
                    double s = v[r];
#pragma unroll
                    for (int c = r + 1; c < M; ++c) s = fmac_row_nn(s, v[r], nv[c], c);   // (v[r]: U's row, written by the elimination long before)
                    nv[r] = -(s * invd[r]);
                }
#pragma unroll
                for (int r = 0; r < M; ++r) v[r] = -nv[r];
            } else {
#pragma unroll
                for (int r = M - 1; r >= 0; --r) {
                    double s = v[r];
#pragma unroll
                    for (int c = r + 1; c < M; ++c) s = fma(-readlane_f64(v[r], c), v[c], s);
                    v[r] = s * invd[r];
                }
            }
#pragma unroll
            for (int a = 0; a < M; ++a) s3_k[a * LK] = -v[a];
        }
        DPILQR_LDS_FENCE();
        if constexpr (PAD) {   // the real block only, in the real layout (n_rec may be odd: 8-byte stores)
            double* Kt = Kout + (gslot * T + t) * m_rec * n_rec;
            double* dt_ = dout + (gslot * T + t) * m_rec;
            const double* krow = sK + min(lane, N - 1);
            double kv[M];
#pragma unroll
            for (int a = 0; a < M; ++a) kv[a] = krow[a * LK];
#pragma unroll
            for (int a = 0; a < M; ++a)   // row by row: n_rec <= 24 lanes each, no division by the run-time row length
                if (a < m_rec && lane < n_rec) store_f64_nt(Kt + a * n_rec + lane, kv[a]);
            if (lane < m_rec) store_f64_nt(dt_ + lane, sK[lane * LK + N]);
        } else {
            double* Kt = Kout + (gslot * T + t) * M * N;
            double* dt_ = dout + (gslot * T + t) * M;
#pragma unroll
            for (int q = 0; q < K_ROUNDS; ++q) store_v2d_nt(Kt + k_out[q], *reinterpret_cast<const v2d*>(sK + k_in[q]));
            store_f64_nt(dt_ + d_idx, sK[d_idx * LK + N]);
        }
        DPILQR_LDS_FENCE();
        MPHASE(3)
        }
        {
        const int lane = phase_lane<REMAT>(lane0);
#include "riccati_mfma_lane.inc"

        __builtin_amdgcn_s_setprio(0);
        if constexpr (C::TEAM_TAIL) {
            // S4 .. S6 by row tile: this wavefront the first, its helper the second (riccati_mfma_team_tail.inc)
            wave_barrier_all();       // [K | d] is in place for the helper
            constexpr int IT = 0;
#include "riccati_mfma_team_tail.inc"
            if (t == 0) wave_barrier_all();   // (the other steps: the barrier at the top of the next step)
        } else {
        // ---- S4: T3^T[c][i] = sum_a Q_uu[a][c] K[a][i]
        // One row tile of controls (m <= 16): T3^T never goes through LDS.  The fp64 MFMA's output layout -- lane (g, c) holds rows
        // g + 4 v of column c -- IS its A-operand layout for the reduction rows 4 v .. 4 v + 3, so S4's accumulator register v of
        // column tile jt is S5's A operand of reduction step v for row tile jt of a1: same products, same order, one store pass,
        // one LDS round trip and six operand loads less per step.
        // (Not with three wavefronts per SIMD: at 168 registers the two tiles held across the phase boundary cost two spilled
        // registers; the variants for few live items -- the latency-bound ones -- have 256.)
        constexpr bool T3R = (T_M == 1) && (T_N == T_NP) && (WAVES != 12);
        v4d t3[T_M][T_N];
        zero_tiles(t3);
        mfma_product<T_M, T_N, MK, LG, LK>(pGuu, pK, t3);
        if constexpr (!T3R) {
#pragma unroll
            for (int it = 0; it < T_M; ++it)
#pragma unroll
                for (int jt = 0; jt < T_N; ++jt)
                    for_rows(16 * it, 0, M, g, colN[jt], [&](int v, int r) { dT3[16 * jt + r * N] = t3[it][jt][v]; });
            DPILQR_LDS_FENCE();
        }
        MPHASE(4)

        // ---- S5: a1 = T3 [K|d] ; a2 = [K|d]^T [Q_ux|Q_u] ; V = ((Q + a1) + a2) + a2^T   (rows < n, cols <= n)
        {
            double vb[T_NP][T_NP][4];
            v4d a1[T_NP][T_NP], a2[T_NP][T_NP];
            zero_tiles(a1);
            zero_tiles(a2);
            if constexpr (T3R) {
#pragma unroll
                for (int ks = 0; ks < MK / 4; ++ks) {
                    double b[T_NP];
#pragma unroll
                    for (int jt = 0; jt < T_NP; ++jt) b[jt] = pK[ks * 4 * LK + 16 * jt];
#pragma unroll
                    for (int it = 0; it < T_NP; ++it)
#pragma unroll
                        for (int jt = 0; jt < T_NP; ++jt) a1[it][jt] = mfma_f64(t3[0][it][ks], b[jt], a1[it][jt]);
                }
            } else {
                mfma_product<T_NP, T_NP, MK, N, LK>(pT3, pK, a1);
            }
            mfma_product<T_NP, T_NP, MK, LK, LG>(pK, pGux, a2);
#pragma unroll
            for (int it = 0; it < T_NP; ++it)
#pragma unroll
                for (int jt = 0; jt < T_NP; ++jt)
                    for_rows(16 * it, 0, NP, g, colN[jt], [&](int v, int r) { dMt[16 * jt + r * LM] = a2[it][jt][v]; });
            DPILQR_LDS_FENCE();
#pragma unroll
            for (int it = 0; it < T_NP; ++it)
#pragma unroll
                for (int jt = 0; jt < T_NP; ++jt) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) vb[it][jt][v] = 0.0;
                    for_rows_p(16 * it, 0, N, g, colLE[jt], [&](int v, int r, bool ok) {   // (stored under the same predicate)
                        const double q = *(ok ? dQ + 16 * jt + r * LQ : sP), mt = *(ok ? dMtT + 16 * jt * LM + r : sP);
                        vb[it][jt][v] = ((q + a1[it][jt][v]) + a2[it][jt][v]) + mt;
                    });
                }
            DPILQR_LDS_FENCE();
#pragma unroll
            for (int it = 0; it < T_NP; ++it)
#pragma unroll
                for (int jt = 0; jt < T_NP; ++jt)
                    for_rows(16 * it, 0, N, g, colLE[jt], [&](int v, int r) { dP[16 * jt + r * LP] = vb[it][jt][v]; });   // V over [K|d], T3^T
            DPILQR_LDS_FENCE();
            // ---- S6: P <- (V + V^T)/2 ; p <- V[:, n]
#pragma unroll
            for (int it = 0; it < T_NP; ++it)
#pragma unroll
                for (int jt = 0; jt < T_NP; ++jt) {
                    for_rows_p(16 * it, 0, N, g, colN[jt], [&](int v, int r, bool ok) {
                        vb[it][jt][v] = 0.5 * (vb[it][jt][v] + *(ok ? dPt + 16 * jt * LP + r : sP));
                    });
                }
            DPILQR_LDS_FENCE();
#pragma unroll
            for (int it = 0; it < T_NP; ++it)
#pragma unroll
                for (int jt = 0; jt < T_NP; ++jt)
                    for_rows(16 * it, 0, N, g, colN[jt], [&](int v, int r) { dP[16 * jt + r * LP] = vb[it][jt][v]; });   // p is in place
        }
        }   // !TEAM_TAIL
        DPILQR_LDS_FENCE();
        MPHASE(5)
        }
    }
    if (singular && sing && lane0 == 0) singular[b] = 1;
    if (stamps && lane0 == 0) {
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

template <int N, int M, int WAVES, int NS, int NC, bool FUSED = false>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_riccati_mfma(
    int B, int T, const double* __restrict__ tiles, const double* __restrict__ mu_arr, double* __restrict__ Kout,
    double* __restrict__ dout, int32_t* __restrict__ singular, const int32_t* __restrict__ items,
    const int32_t* __restrict__ n_items, int gains_by_item, int n_cus, FusedArgs F) {
    riccati_mfma_sweep<N, M, WAVES, NS, NC, (FUSED ? 1 : 0)>(B, T, tiles, mu_arr, Kout, dout, singular, items, n_items, gains_by_item, n_cus, F);
}

// The fused sweeps with a helper wavefront per item (HELP above): 512-thread workgroups, wavefronts 0..3 sweep four items (one
// per SIMD), wavefronts 4..7 are their helpers.  FUSED: 1 the DoubleInt4D form, 2 the general four-state form.
template <int N, int M, int FUSED>
__global__ __launch_bounds__(512, 2) void k_riccati_mfma_team(
    int B, int T, const double* __restrict__ mu_arr, double* __restrict__ Kout, double* __restrict__ dout,
    int32_t* __restrict__ singular, const int32_t* __restrict__ items, const int32_t* __restrict__ n_items, int gains_by_item,
    int n_cus, FusedArgs F) {
    riccati_mfma_sweep<N, M, 4, 4, 2, FUSED, false, true>(B, T, nullptr, mu_arr, Kout, dout, singular, items, n_items, gains_by_item,
                                                          n_cus, F);
}

// The dense record-fed sweep for cluster sizes without an instantiation of their own: records and gains of size (n_rec, m_rec),
// padded into (N, M) while loading (PAD above).
template <int N, int M, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_riccati_mfma_pad(
    int B, int T, const double* __restrict__ tiles, const double* __restrict__ mu_arr, double* __restrict__ Kout,
    double* __restrict__ dout, int32_t* __restrict__ singular, const int32_t* __restrict__ items,
    const int32_t* __restrict__ n_items, int gains_by_item, int n_cus, int n_rec, int m_rec) {
    riccati_mfma_sweep<N, M, WAVES, 0, 0, 0, true>(B, T, tiles, mu_arr, Kout, dout, singular, items, n_items, gains_by_item, n_cus,
                                                   FusedArgs{}, n_rec, m_rec);
}

// In-sweep production (PNS above): the record-free sweep for clusters of the six-state family and of CarDynamics3D, (n_rec, m_rec)
// = k (PNS, PNC) padded into (N, M) (PAD), or of exactly that size.
template <int N, int M, int WAVES, int PNS, bool PAD>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_riccati_mfma_inprod(
    int B, int T, const double* __restrict__ mu_arr, double* __restrict__ Kout, double* __restrict__ dout,
    int32_t* __restrict__ singular, const int32_t* __restrict__ items, const int32_t* __restrict__ n_items, int gains_by_item,
    int n_cus, FusedArgs F, int n_rec, int m_rec) {
    riccati_mfma_sweep<N, M, WAVES, 0, 0, 0, PAD, false, PNS>(B, T, nullptr, mu_arr, Kout, dout, singular, items, n_items,
                                                              gains_by_item, n_cus, F, PAD ? n_rec : N, PAD ? m_rec : M);
}

// The record-free sweep's general form for the four-state family (FUSED = 2 above): at most five agents of one model --
// DoubleIntDynamics4D or UnicycleDynamics4D -- with per-agent, per-item weights.
template <int N, int M, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_riccati_mfma_general(
    int B, int T, const double* __restrict__ mu_arr, double* __restrict__ Kout, double* __restrict__ dout,
    int32_t* __restrict__ singular, const int32_t* __restrict__ items, const int32_t* __restrict__ n_items, int gains_by_item,
    int n_cus, FusedArgs F) {
    riccati_mfma_sweep<N, M, WAVES, 4, 2, 2>(B, T, nullptr, mu_arr, Kout, dout, singular, items, n_items, gains_by_item, n_cus, F);
}

}  // namespace dpilqr
