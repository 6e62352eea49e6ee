// riccati_wg.hpp -- K2 for the larger clusters (n_x 24 .. 60): the Riccati backward sweep with ONE WORKGROUP of four
// wavefronts per sub-problem (ilqrSolver._backward_pass, control.py:116-148).
//
// riccati_mfma.hpp gives every sub-problem one wavefront and 13.5 KB of LDS; that stops at n_x = 20.  A 15-agent
// unicycle cluster (cfg3) has n_x = 60, n_u = 30, and the work of one step (a 30 x 30 pivoted LU with 61 right-hand
// sides, 0.27 M dense FMAs) is enough for four wavefronts.  Same recursion, same association order as the reference
// (see riccati.hpp for the equations):
//
//   S0  this step's diagonal blocks of [A|B] registers -> LDS (requested during the previous step); request l-values
//   S1  [T1;T2 | A^T p;B^T p] = [A|B]^T [P|p]  (+ mu B^T on the B rows): block-diagonal [A|B] (the library's own
//       tiles, see riccati_mfma.hpp), NS terms per output, work items (agent, 2 columns) dealt to the 256 lanes
//   S2  [T1;T2][A|B] + l-values -> Q_xx, Q_ux, Q_uu: work items (agent, 2 rows)
//   S3  LU with partial pivoting, in registers: EVERY wavefront factorises Q_uu (lanes 0..m-1 hold its columns) and
//       carries its own 64-m of the n+1 right-hand sides, so no pivot or multiplier ever crosses a wavefront
//   S4  T3^T = Q_uu-contracted K            fp64 MFMA tiles dealt round-robin to the wavefronts
//   S5  a1 = T3 [K|d], a2 = [K|d]^T [Q_ux|Q_u], V = ((Q + a1) + a2) + a2^T          likewise
//   S6  P <- (V + V^T)/2
//
// The kernel is bound by latency, not by a pipe (one step at n_x = 60: 73 k cycles, of which the matrix pipe is busy
// 4 k), so what it needs is several sub-problems per CU, and what decides that is LDS and registers.  Three regions:
//   P  n x (n+1):  [P|p] -> T1 = A^T P in place (an S1 work item reads and writes the same rows of its two columns)
//                  -> [Q_xx|Q_x] in place (an S2 work item reads and writes the same block of its rows) -> V -> [P|p]
//   G  m x (m + n + 1):  T2 = B^T (P + mu I) at the place of Q_ux -> [Q_uu | Q_ux | Q_u] (S2, in place)
//   K  [K|d] + T3^T (S3 .. S5 operands); a2 (S5 epilogue, read transposed by S6) takes G and K over once both are dead
// plus the step's diagonal blocks of [A|B] (k n_s (n_s + n_c) doubles; the dense record rows are never staged).
// 74 KB at n_x = 60 (two sub-problems per CU; 147 KB and one before), 48 KB at 42 .. 48 (three), 35 KB at 36 (four); the register
// budget follows (__launch_bounds__).  Phases are separated by workgroup barriers (s_barrier behind an LDS-only wait:
// the global prefetches stay in flight).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "riccati_mfma.hpp"

namespace dpilqr {

constexpr int kWgThreads = 256;

template <int N, int M, int NS, int NC, bool FUSED = false>
struct WgCfg {
    static constexpr int NM = N + M, NP = N + 1, MK = round_up(M, 4), KA = N / NS, NSC = NS + NC;
    static constexpr int LP = round_up(NP, 2), LQ = LP, LK = LP, LM = N, KROWS = MK + 2;
    static constexpr int MO = round_up(M, 2);                  // column of Q_ux inside a row of G (16-byte aligned)
    static constexpr int LG = round_up(MO + NP, 2);            // [Q_uu | pad | Q_ux | Q_u]
    static constexpr int NSCP = round_up(NSC, 2);              // one row of an agent's [A_blk | B_blk]
    static constexpr int T_NP = (NP + 15) / 16, T_N = (N + 15) / 16, T_M = (M + 15) / 16;
    static constexpr int szP = round_up(N * LP, 2);
    static constexpr int szG = round_up(MK * LG, 2);
    // four column tiles (n_x >= 48): wavefront w owns column tile w of T3^T and row tile w of a1, a2, and the fp64 MFMA's
    // output layout (lane (g, c) holds rows g + 4 v) is its A-operand layout for the reduction rows 4 v .. 4 v + 3, so
    // T3^T goes from S4's accumulators straight into S5's products: no LDS for it (15 KB at n_x = 60: the difference
    // between one and two sub-problems per CU), no barrier between S4 and S5
    static constexpr bool T3REG = (T_NP == 4);
    static constexpr int szKT3 = KROWS * LK + (T3REG ? 0 : MK * N) + 16;
    // a2 (n+1 x n, S5 epilogue -> S6) starts at G and runs on into the [K|d] region: both are dead by then
    static constexpr int szK = round_up(szG + szKT3 >= NP * LM ? szKT3 : NP * LM - szG, 2);
    static constexpr int szAB = round_up(KA * NS * NSCP, 2);
    // S3 by blocks (gj_blocked below) for m = 13 .. 16 (one row tile: 16 registers of accumulators) and from m = 22 on (where
    // the register budget of two sub-problems per CU holds both it and the fall-back); at m = 17 .. 21 it needs two sub-problems
    // per CU where the register LU runs with three or four, and measures slower (m = 18: 4.4 -> 5.6 ms per 2048 items) or equal
    // (m = 20, 21); m = 22 (eleven unicycles) gains 5 .. 9 % (-DDPILQR_GJ_ALL builds, profiles/r03_wg_gj_all.txt).  32 x 4 doubles per wavefront to turn a panel's columns
    // into rows
#ifdef DPILQR_GJ_ALL   // A/B builds
    static constexpr bool GJ = (M >= 13);
#else
    static constexpr bool GJ = (M >= 22) || (M >= 13 && M <= 16);
#endif
    // ... which take the place of the [A|B] blocks where those are large enough (dead between S2 and the next S0)
    static constexpr bool PAN_IN_AB = GJ && szAB >= 4 * 128;
    static constexpr int szPan = (GJ && !PAN_IN_AB) ? 4 * 128 : 0;
    // FUSED (no tile records, see the kernel): w_ref (Q + Q^T), w_ref-less (R + R^T) of every agent, once per sub-problem.
    // The step's plugin data -- x, x - x_f, u, pair gradients and Hessians, their per-agent sums -- live in rows < m of the
    // [K|d] region, which is dead from the end of S6 to S3
    static constexpr int NPR = KA * (KA - 1) / 2;
    static constexpr int szW = FUSED ? round_up(KA * (NS * NS + NC * NC), 2) : 0;
    // (offsets from the start of [K|d]; where T3^T has an LDS region of its own -- dead until S4 -- x, x - x_f, u go there)
    static constexpr int oFg = 0, oFh = oFg + 3 * NPR, oFd = oFh + 9 * NPR, oFs = oFd + 9 * KA, oFend0 = oFs + 3 * KA;
    static constexpr bool XU_IN_T3 = !T3REG;
    static constexpr int oFx = XU_IN_T3 ? KROWS * LK : oFend0, oFe = oFx + N, oFu = oFe + N;
    static constexpr int oFend = XU_IN_T3 ? oFend0 : oFu + M;
    static constexpr bool fused_fits = oFend <= M * LK && (!XU_IN_T3 || 2 * N + M <= MK * N);
    static constexpr int oP = 0, oG = oP + szP, oK = oG + szG, oT3 = oK + KROWS * LK, oAB = oK + szK, oPan = oAB + szAB,
                         oW = oPan + szPan, oEnd = oW + szW;
    static constexpr int total = round_up(oEnd + 64, 2);   // + store target of idle lanes, wrapped tile reads
    static constexpr bool ALA = (NS % 2 == 0);                   // an agent's column block starts at an even offset
    static constexpr bool AL = ALA && (NC % 2 == 0);
    static constexpr int CG = (NP + 1) / 2;                  // S1: column groups of 2 over [P|p]
    static constexpr int NI1 = KA * CG, R1R = (NI1 + kWgThreads - 1) / kWgThreads;
    static constexpr int RPL = 2, RG = (NM + RPL - 1) / RPL;  // S2: row groups
    static constexpr int NI2 = KA * RG, R2R = (NI2 + kWgThreads - 1) / kWgThreads;
    static constexpr int ABN = KA * NS * NSC;                                  // doubles in the diagonal blocks of [A|B]
    static constexpr int ABR = (ABN + kWgThreads - 1) / kWgThreads;            // ... a lane stages
    static constexpr int NT4 = T_M * T_N, TPW4 = (NT4 + 3) / 4;              // S4 tiles, per wavefront
    static constexpr int NT5 = T_NP * T_NP, TPW5 = (NT5 + 3) / 4;            // S5 tiles, per wavefront
    // sub-problems per CU: what the LDS allows, capped by what the registers allow without spilling (512 / that per lane;
    // the LU alone holds 4 m of them, the S1 / S2 accumulators 4 (n_s + n_c)).  A spilling build is far slower than a
    // build with one sub-problem fewer per CU (measured), so the caps follow the compiler's spill-free register counts
    static constexpr int kLdsFit = (160 * 1024) / (total * 8);
#ifdef DPILQR_GJ_ALL
    static constexpr int kRegFit = NS >= 12 ? 2 : (NS >= 6 ? (M <= 16 ? 3 : 2) : (M <= 16 ? 4 : 2));
#else
    static constexpr int kRegFit = NS >= 12 ? 2 : (NS >= 6 ? (M <= 21 ? 3 : 2) : (M <= 18 ? 4 : (M <= 21 ? 3 : 2)));
#endif
#ifdef DPILQR_WG_OCC   // A/B builds
    static constexpr int OCC = kLdsFit < DPILQR_WG_OCC ? (kLdsFit < 1 ? 1 : kLdsFit) : DPILQR_WG_OCC;
#else
    static constexpr int OCC = kLdsFit < 1 ? 1 : (kLdsFit < kRegFit ? kLdsFit : kRegFit);
#endif
    static constexpr bool supported = (N % NS == 0) && (M == KA * NC) && (N % 2 == 0) && (M <= 32) && (64 - M > 0) &&
                                      (4 * (64 - M) >= NP) && (total * 8 <= 160 * 1024) &&
                                      (!FUSED || (fused_fits && 12 * KA <= kWgThreads && 64 + NPR <= kWgThreads));
};

// Pins a phase's accumulators at this point of the program: the multiply-adds that produce them are issued before it, the
// LDS loads behind it after it.  Without it the instruction selector sinks all multiply-adds of an unrolled reduction
// behind all of its operand loads (24 n_s registers of loaded rows), and registers decide how many sub-problems share a CU.
template <int K>
__device__ __forceinline__ void pin_regs(double (&a)[K]) {
#pragma unroll
    for (int i = 0; i < K; ++i) asm volatile("" : "+v"(a[i]));
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int LEN, bool VEC>
__device__ __forceinline__ void ld_row(const double* p, double* out) {   // LEN doubles; VEC: p is 16-byte aligned
    if constexpr (VEC) {
#pragma unroll
        for (int q = 0; q < LEN / 2; ++q) {
            const v2d v = *reinterpret_cast<const v2d*>(p + 2 * q);
            out[2 * q] = v.x; out[2 * q + 1] = v.y;
        }
        if constexpr (LEN % 2 == 1) out[LEN - 1] = p[LEN - 1];
    } else {
#pragma unroll
        for (int q = 0; q < LEN; ++q) out[q] = p[q];
    }
}
template <int LEN, bool VEC>
__device__ __forceinline__ void st_row(double* p, const double* v) {
    if constexpr (VEC) {
#pragma unroll
        for (int q = 0; q < LEN / 2; ++q) *reinterpret_cast<v2d*>(p + 2 * q) = v2d{v[2 * q], v[2 * q + 1]};
        if constexpr (LEN % 2 == 1) p[LEN - 1] = v[LEN - 1];
    } else {
#pragma unroll
        for (int q = 0; q < LEN; ++q) p[q] = v[q];
    }
}


// FUSED S0, one lane per agent: MultiDynamicalModel.linearize's per-agent call (dynamics.py:173-186) -> the agent's rows of
// [A_blk | B_blk] in LDS.  (Measured out of line as well -- the run-time model switch carries the sincos / tan expansions of
// several models -- with no effect on the kernel's register allocation and a per cent or two of launch time lost to the call.)
template <int NS, int NC, int NSCP>
__device__ __forceinline__ void wg_linearize_agent(int model, const double* __restrict__ sx, const double* __restrict__ su,
                                                             double dt, double* __restrict__ sab) {
    double x[NS], u[NC], A[NS * NS], Bm[NS * NC];
#pragma unroll
    for (int i = 0; i < NS; ++i) x[i] = sx[i];
#pragma unroll
    for (int i = 0; i < NC; ++i) u[i] = su[i];
    linearize_rt<NS>(model, x, u, dt, A, Bm);
#pragma unroll
    for (int l = 0; l < NS; ++l) {
#pragma unroll
        for (int c = 0; c < NS; ++c) sab[l * NSCP + c] = A[l * NS + c];
#pragma unroll
        for (int c = 0; c < NC; ++c) sab[l * NSCP + NS + c] = Bm[l * NC + c];
    }
}

// S3's fall-back (and, where gj_blocked is not used, S3 itself): LU with LAPACK-order partial pivoting in registers.  Lanes
// 0..M-1 hold Q_uu's columns, lanes M..63 this wavefront's right-hand sides; [K|d] columns of this wavefront -> sK.
// Returns bit 0: an exactly zero pivot; bit 1: the search-free elimination ran (strictly column-dominant Q_uu); bit 2: partial
// pivoting moved a row.  NOT inlined (see the call).
template <int M, int N, int NP, int MO, int LG, int LK, int RW>
__device__ __attribute__((noinline)) int lu_fallback_wg(const double* __restrict__ sG, double* __restrict__ sK, int wave, int lane) {
    int sing = 0;
    const int s3_q = RW * wave + (lane - M);
    const bool s3_rhs = lane >= M && s3_q < NP;
    const int s3_col = lane < M ? lane : MO + min(max(s3_q, 0), N);
    double v[M], invd[1];   // the reciprocal pivots are recomputed in the substitution: 2 m registers less
#pragma unroll
    for (int r = 0; r < M; ++r) v[r] = sG[r * LG + s3_col];
    // strictly column-dominant Q_uu: no row moves, the elimination runs without the pivot search
    // (see S3 of k_riccati_mfma for the argument and the margin)
    double colsum = 0.0;
#pragma unroll
    for (int r = 0; r < M; ++r) colsum = colsum + fabs(v[r]);
    const double dg = fabs(sG[min(lane, M - 1) * (LG + 1)]);
    const bool dominant = lane >= M || dg * (1.0 - 0x1p-20) > colsum - dg;
    const bool no_swaps = __builtin_amdgcn_ballot_w64(!dominant) == 0ull;
    int n_swaps = 0;
    if (no_swaps) lu_eliminate<false, M, false, false>(v, invd, sing);
    else lu_eliminate<true, M, false, false>(v, invd, sing, &n_swaps);
#pragma unroll
    for (int r = M - 1; r >= 0; --r) {
        double s = v[r];
#pragma unroll
        for (int c = r + 1; c < M; ++c) s = fma(-readlane_f64(v[r], c), v[c], s);
        const double pv = readlane_f64(v[r], r);   // U's diagonal: the pivot of step r
        double inv = __builtin_amdgcn_rcp(pv);
        inv = fma(fma(-pv, inv, 1.0), inv, inv);
        inv = fma(fma(-pv, inv, 1.0), inv, inv);
        v[r] = s * inv;
    }
    if (s3_rhs) {
#pragma unroll
        for (int a = 0; a < M; ++a) sK[a * LK + s3_q] = -v[a];
    }
    return (sing ? 1 : 0) | (no_swaps ? 2 : 0) | (n_swaps > 0 ? 4 : 0);
}



// S3 by blocks: [K | d] = -Q_uu^-1 [Q_ux | Q_u] by Gauss-Jordan elimination on 4-column panels, the row operations of a panel
// applied to everything to its right as ONE fp64 MFMA per 16 x 16 tile.
//
// The register LU (lu_eliminate) is a chain of m^2 / 2 broadcast-multiply-add steps, three to four instructions each --
// 50 k cycles at m = 30, 60 % of a sweep step.  Here a wavefront keeps [Q_uu | its 16 right-hand sides] in the MFMA's
// output layout (lane (g, c) holds rows 16 it + g + 4 v of column 16 jt + c).  In that layout the four rows of a panel,
// K .. K + 3 with K = 16 it + 4 v, are ONE register across the lane groups g = 0 .. 3 -- which is exactly the B operand
// of v_mfma_f64_16x16x4 (lane (g, c) supplies row g of the 4-row reduction block).  So a panel step is:
//   1. the panel's four columns go through LDS into "row layout" (every 16-lane row holds matrix rows c and 16 + c);
//   2. Gauss-Jordan on those four columns in registers, multipliers in-lane, pivot-row entries by DPP row broadcast,
//      applied at the same time to the 4 columns of the identity that belong to the panel's rows: that yields W (m x 4),
//      the panel's part of the row-operation matrix (pivot rows scaled so that the diagonal becomes 1);
//   3. [everything right of the panel] = [the same with the panel's rows cleared] + W * [the panel's rows]: one MFMA per tile,
//      A operand = W (a lane picks column g of its rows), B operand = the accumulator register that holds the panel's rows.
// After the last panel Q_uu has become I and the right-hand sides hold the solution; no substitution pass.
//
// Pivoting: threshold pivoting that prefers the diagonal.  The diagonal entry is the pivot as long as no entry below it in its
// column is more than 8 times larger (the threshold rule of sparse LU with a diagonal preference -- UMFPACK / MA48 use 0.1
// there, KLU 0.001; element growth per step is bounded by 1 + 8 instead of partial pivoting's 2); where the rule fails -- 7 % of
// the steps of 15-unicycle iterates have such a column (scripts/threshold_study.py) -- the largest entry below the diagonal comes
// up (dgetf2's choice among those rows): the two rows change places in the panel, in W's finished columns and, before the
// panel's update, in the tiles it touches (see the swap block below).  Round 3 declined the whole step instead and ran a
// 50 k-cycle register LU all four wavefronts waited for; a looser threshold declined less often but the end-to-end envelope
// (oracle/parity.py) saw the price (worst cost error 2.6 / 4.7 / 7.5 / 12.9 ensemble spreads at 8 / 16 / 32 / 64,
// profiles/r03_gj_threshold_tails.txt), so 8 stays.  With the swaps, on 2048 real iterates (profiles/r04_gj_row_swaps.txt): the
// record-fed sweep 8.39 -> 7.51 ms (ten quadcopters, T = 75) and 11.27 -> 9.73 ms (fifteen unicycles, T = 100), the fused one
// 9.04 -> 8.73 and 12.27 -> 11.53.  (DPILQR_GJ_THRESHOLD, DPILQR_GJ_NO_SWAP: A/B builds.)
// Round 5, the same audit with the swaps in place per threshold 1 (dgetf2's rule) .. 64 (profiles/r05_gj_threshold_tails.txt): the tails at
// the sizes this elimination serves do not depend on the threshold any more (the largest, 3.52 spreads, under true partial pivoting): they
// are the solves' chaos, not element growth.  Threshold 1 costs 7 % of a sweep, 2 3 %, 4 1 %; 64 gains 0.5 %.
// Q_uu = R + B^T (P + mu I) B is symmetric with a heavy diagonal; it is NOT always positive definite away from a minimum (half
// of the steps of a fresh 15-unicycle iterate have a negative pivot), which is why the rule looks at magnitudes.  Only a pivot
// that is zero or not finite after the search (a singular or poisoned Q_uu) makes the caller run the register LU with
// LAPACK-order partial pivoting (lu_eliminate<true>), which reports the singularity; nothing has been written by then.  Every
// wavefront factorises the same Q_uu with the same instructions, so all of them take the same decisions.
#ifndef DPILQR_GJ_THRESHOLD
#define DPILQR_GJ_THRESHOLD 8
#endif
constexpr double kGjThreshold = 1.0 / DPILQR_GJ_THRESHOLD;

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N_, typename F>
__device__ __forceinline__ void static_for(F& f) { static_for_impl(f, std::make_integer_sequence<int, N_>{}); }

template <int M, int MO, int LG, int LK, int NP>
__device__ __forceinline__ bool gj_blocked(const double* __restrict__ sG, double* __restrict__ sK, double* __restrict__ sPan,
                                           int wave, int lane) {
    constexpr int MK = round_up(M, 4), RT = (M + 15) / 16, CT = RT + 1, NPAN = MK / 4;
    const int g = lane >> 4, c = lane & 15;
    v4d acc[RT][CT];
#pragma unroll
    for (int it = 0; it < RT; ++it)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = 16 * it + g + 4 * v;
#pragma unroll
            for (int jt = 0; jt < RT; ++jt) {
                const int col = 16 * jt + c;
                const double x = sG[min(row, M - 1) * LG + min(col, M - 1)];
                acc[it][jt][v] = (row < M && col < M) ? x : (row == col ? 1.0 : 0.0);   // identity beyond m
            }
            const int q = 16 * wave + c;
            const double y = sG[min(row, M - 1) * LG + MO + min(q, NP - 1)];
            acc[it][RT][v] = (row < M && q < NP) ? y : 0.0;
        }
    bool bad = false;
    // (the panels by instantiation, not by `#pragma unroll`: with the row swaps the loop outgrew the size up to which the pragma is
    // honoured, and a panel index that is not a constant puts `acc` into scratch memory -- the sweep ran 3.8 times slower)
    auto panel = [&](auto p_tag) __attribute__((always_inline)) {
        constexpr int p = decltype(p_tag)::value;
        constexpr int Kp = 4 * p, itp = Kp / 16, vp = (Kp % 16) / 4, cp = Kp % 16;
        // 1. the panel's columns -> row layout
        if (c >= cp && c < cp + 4) {
#pragma unroll
            for (int it = 0; it < RT; ++it)
#pragma unroll
                for (int v = 0; v < 4; ++v) sPan[(16 * it + g + 4 * v) * 4 + (c - cp)] = acc[it][itp][v];
        }
        DPILQR_LDS_FENCE();
        // 2. Gauss-Jordan on the panel, W alongside -- first WITHOUT the row swaps, the threshold rule only noted per lane: one
        // ballot per panel instead of one per pivot on the common path; a panel in which the rule failed somewhere (a few per
        // cent of the steps have one) is done again from its copy in LDS with the swaps (nothing outside pan / W has changed)
        double W[RT][4];
        int sw_r[4] = {-1, -1, -1, -1};     // row that changed places with row Kp + j, if any
        auto gj_panel = [&](auto swaps_tag, bool& bad_p) __attribute__((always_inline)) -> bool {
            constexpr bool SWAPS = decltype(swaps_tag)::value;
            bool viol_any = false;
            double pan[RT][4];
#pragma unroll
            for (int it = 0; it < RT; ++it) {
                const v2d x = *reinterpret_cast<const v2d*>(sPan + (16 * it + c) * 4);
                const v2d y = *reinterpret_cast<const v2d*>(sPan + (16 * it + c) * 4 + 2);
                pan[it][0] = x.x; pan[it][1] = x.y; pan[it][2] = y.x; pan[it][3] = y.y;
#pragma unroll
                for (int j = 0; j < 4; ++j) W[it][j] = (16 * it + c == Kp + j) ? 1.0 : 0.0;
            }
            DPILQR_LDS_FENCE();
            // (the panel itself)
            double invs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int L = cp + j;
                double pv = mov_row(pan[itp][j], L);
                if constexpr (!SWAPS) {   // the threshold rule, noted only (one decision per panel, see below)
#pragma unroll
                    for (int it = 0; it < RT; ++it)
                        viol_any = viol_any || (16 * it + c > Kp + j && kGjThreshold * fabs(pan[it][j]) > fabs(pv));
                } else {   // the threshold rule; where it fails, the largest entry below comes up: rows swapped
                    const double pv0 = pv;
                    bool viol = false;
#pragma unroll
                    for (int it = 0; it < RT; ++it)
                        viol = viol || (16 * it + c > Kp + j && kGjThreshold * fabs(pan[it][j]) > fabs(pv0));
                    if (__builtin_amdgcn_ballot_w64(viol) != 0ull) {
                        // first largest |entry| of column Kp + j below the diagonal (dgetf2's idamax over those rows): every 16-lane row
                        // of the wavefront holds the same copy of the panel, so the reduction stays inside a row
                        double best = 0.0;
                        int brow = 0;
#pragma unroll
                        for (int it = 0; it < RT; ++it) {
                            const double av = fabs(pan[it][j]);
                            const bool take = (16 * it + c > Kp + j) && av > best;
                            best = take ? av : best;
                            brow = take ? 16 * it + c : brow;
                        }
#pragma unroll
                        for (int off = 1; off < 16; off <<= 1) {
                            const double ob = __shfl_xor(best, off, 16);
                            const int orow = __shfl_xor(brow, off, 16);
                            const bool take = ob > best || (ob == best && ob > 0.0 && orow < brow);
                            best = take ? ob : best;
                            brow = take ? orow : brow;
                        }
                        const int r = __builtin_amdgcn_readfirstlane(brow);
                        if (r > Kp + j) {
                            // rows Kp + j and r change places: in the panel (all four columns) and in W's finished columns (the multipliers
                            // are attributes of the rows; the unit entries of the columns still to come belong to the positions) now; in
                            // the tiles the panel's update will touch -- its B operand is then the rows that are pivots now -- before
                            // that update (sw_r)
                            const int itB = r >> 4, cB = r & 15;
                            const int cA = L;
                            const int srcB = (lane & 48) | cB, srcA = (lane & 48) | cA;
                            auto swap_row_layout = [&](double (&x)[RT][4], int col) {
                                double vb = x[0][col];
#pragma unroll
                                for (int it = 1; it < RT; ++it) vb = (it == itB) ? x[it][col] : vb;
                                const double fromB = __shfl(vb, srcB), fromA = __shfl(x[itp][col], srcA);
#pragma unroll
                                for (int it = 0; it < RT; ++it) x[it][col] = (it == itB && c == cB) ? fromA : x[it][col];
                                x[itp][col] = (c == cA) ? fromB : x[itp][col];
                            };
#pragma unroll
                            for (int col = 0; col < 4; ++col) swap_row_layout(pan, col);
#pragma unroll
                            for (int jp = 0; jp < j; ++jp) swap_row_layout(W, jp);
                            sw_r[j] = r;
                            pv = mov_row(pan[itp][j], L);
                        }
                    }
                }
                double inv = __builtin_amdgcn_rcp(pv);
                inv = fma(fma(-pv, inv, 1.0), inv, inv);
                inv = fma(fma(-pv, inv, 1.0), inv, inv);
                invs[j] = inv;
                bad_p = bad_p || !(fabs(pv) > 0.0) || !(fabs(pv) < 1.0e300);
                const double ninv = -inv;
                double l[RT];
#pragma unroll
                for (int it = 0; it < RT; ++it) {
                    const int row = 16 * it + c;
                    const double a = pan[it][j];
                    l[it] = (row != Kp + j) ? a * ninv : 0.0;
                }
                // The DPP operand of these updates is the pivot row's entry, pan[itp][jj] / W[itp][jp] read from lane L.  The row
                // tile that holds the pivot row goes LAST: every other tile's update then reads a register that was last written a
                // whole pivot stage ago, and the pivot tile's own update reads the register it overwrites -- no update follows a
                // write of its DPP operand within two instructions, so none needs the s_nop (6 RT per pivot, 290 per step at
                // m = 30).  (fmac_row_nn is a volatile asm: the updates keep this order.)
#pragma unroll
                for (int jj = j + 1; jj < 4; ++jj)
#pragma unroll
                    for (int io = 1; io <= RT; ++io) {
                        const int it = (itp + io) % RT;
                        pan[it][jj] = fmac_row_nn(pan[it][jj], pan[itp][jj], l[it], L);
                    }
#pragma unroll
                for (int jp = 0; jp < j; ++jp)
#pragma unroll
                    for (int io = 1; io <= RT; ++io) {
                        const int it = (itp + io) % RT;
                        W[it][jp] = fmac_row_nn(W[it][jp], W[itp][jp], l[it], L);
                    }
#pragma unroll
                for (int it = 0; it < RT; ++it) W[it][j] = (16 * it + c != Kp + j) ? l[it] : W[it][j];
            }
            {   // the pivot rows, scaled: the diagonal becomes 1
                double sc = 1.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) sc = (c == cp + j) ? invs[j] : sc;
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) W[itp][jp] = W[itp][jp] * sc;
            }
            return viol_any;
        };
#ifdef DPILQR_GJ_CHECK_PER_PIVOT   // (A/B builds: the swap form on every panel -- a ballot per pivot)
        {
            bool bad_slow = false;
            (void)gj_panel(std::true_type{}, bad_slow);
            bad = bad || bad_slow;
        }
#else
        {
            bool bad_fast = false;
            const bool viol = gj_panel(std::false_type{}, bad_fast);
#ifdef DPILQR_GJ_NO_SWAP   // (A/B builds: the round-3 form, which declined the whole step instead)
            bad = bad || bad_fast || viol;
#else
            if (__builtin_amdgcn_ballot_w64(viol) != 0ull) {
                bool bad_slow = false;
                (void)gj_panel(std::true_type{}, bad_slow);
                bad = bad || bad_slow;
            } else {
                bad = bad || bad_fast;
            }
#endif
        }
#endif
#ifndef DPILQR_GJ_NO_SWAP
        // the panel's row swaps, in their order, on the tiles the update is about to touch.  Row Kp + s is register vp of lane
        // group s of row tile itp; its partner r is found at run time (a loop, not unrolled: rare, and four copies of it per panel
        // made the kernel too large for the compiler to keep its arrays in registers)
        if (max(max(sw_r[0], sw_r[1]), max(sw_r[2], sw_r[3])) >= 0) {
#pragma unroll 1
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int r = sidx == 0 ? sw_r[0] : (sidx == 1 ? sw_r[1] : (sidx == 2 ? sw_r[2] : sw_r[3]));
                if (r < 0) continue;
                const int itB = r >> 4, vB = (r & 15) >> 2, gB = r & 3;
#pragma unroll
                for (int jt = 0; jt < CT; ++jt) {
                    if (jt < itp) continue;
                    double y = 0.0;
#pragma unroll
                    for (int it = 0; it < RT; ++it)
#pragma unroll
                        for (int v = 0; v < 4; ++v) y = (it == itB && v == vB) ? acc[it][jt][v] : y;
                    const double x = acc[itp][jt][vp];
                    const double yA = __shfl(y, 16 * gB + c), xB = __shfl(x, 16 * sidx + c);
#pragma unroll
                    for (int it = 0; it < RT; ++it)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            double e = acc[it][jt][v];
                            e = (it == itB && v == vB && g == gB) ? xB : e;
                            e = (it == itp && v == vp && g == sidx) ? yA : e;
                            acc[it][jt][v] = e;
                        }
                }
            }
        }
#endif
        // 3. everything from the panel's tile on: += W * (the panel's rows), the panel's rows themselves replaced
        // (W pinned first: seen through, the selects that built W and this one are merged into control flow -- seven
        // branches per panel)
        double a_op[RT];
#pragma unroll
        for (int it = 0; it < RT; ++it) {
            asm volatile("" : "+v"(W[it][0]), "+v"(W[it][1]), "+v"(W[it][2]), "+v"(W[it][3]));
            double r = W[it][0];
            r = g == 1 ? W[it][1] : r; asm volatile("" : "+v"(r));
            r = g == 2 ? W[it][2] : r; asm volatile("" : "+v"(r));
            r = g == 3 ? W[it][3] : r;
            a_op[it] = r;
        }
        const int jt_first = (Kp + 4) / 16 < RT ? (Kp + 4) / 16 : itp;   // the tile the next panel's columns are in goes first
#pragma unroll
        for (int jo = 0; jo < CT; ++jo) {
            const int jt = jo == 0 ? jt_first : (jo <= jt_first ? jo - 1 : jo);
            if (jt < itp) continue;   // columns left of the panel's tile are done (unit columns, zero in the panel's rows)
            const double b = acc[itp][jt][vp];
            acc[itp][jt][vp] = 0.0;
#pragma unroll
            for (int it = 0; it < RT; ++it) acc[it][jt] = mfma_f64(a_op[it], b, acc[it][jt]);
        }
    };
    static_for<NPAN>(panel);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) return true;
#pragma unroll
    for (int it = 0; it < RT; ++it)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = 16 * it + g + 4 * v, q = 16 * wave + c;
            if (row < M && q < NP) sK[row * LK + q] = -acc[it][RT][v];
        }
    return false;
}

// FUSED: no tile records.  The sweep evaluates MultiDynamicalModel.linearize and GameCost.quadraticize itself, from
// (X[t], U[t]) -- n_x + n_u doubles per step, requested a step ahead -- with the tile producer's own expressions and
// summation orders (tiles.hpp), so the gains are those of the record-fed sweep bit for bit.  A step's S0 becomes: (x, x - x_f,
// u) -> LDS | agents' [A_i|B_i] (one lane per agent) and the pair derivatives (one lane per pair) | their per-agent sums;
// the l-values are formed by the S1 / S2 work items that add them.  Any models of the state family, any per-agent weights.
template <int N, int M, int NS, int NC, bool FUSED = false>
__global__ __launch_bounds__(kWgThreads, (WgCfg<N, M, NS, NC, FUSED>::OCC)) void k_riccati_wg(
    int B, int T, const double* __restrict__ tiles, const double* __restrict__ mu_arr, double* __restrict__ Kout,
    double* __restrict__ dout, int32_t* __restrict__ singular, const int32_t* __restrict__ items,
    const int32_t* __restrict__ n_items, int gains_by_item, FusedArgs F) {
    using C = WgCfg<N, M, NS, NC, FUSED>;
    constexpr int NM = C::NM, NP = C::NP, MK = C::MK, KA = C::KA, NSC = C::NSC, NSCP = C::NSCP, LP = C::LP, LQ = C::LQ;
    constexpr int LG = C::LG, LK = C::LK, LM = C::LM, MO = C::MO, T_NP = C::T_NP, T_N = C::T_N;
    constexpr bool AL = C::AL, ALA = C::ALA;
#ifndef DPILQR_WG_NO_STRUCT4   // (A/B builds)
    constexpr bool STRUCT4 = FUSED && NS == 4 && NC == 2;   // the library's own four-state models: S1 / S2 by the blocks' structure
#else
    constexpr bool STRUCT4 = false;
#endif
    const int slot = blockIdx.x;
    if (slot >= (n_items ? *n_items : B)) return;
    const int b = items ? items[slot] : slot;
    if (b >= B) return;
    const int64_t gslot = gains_by_item ? b : slot;
    const int tid = threadIdx.x, tid_k = tid;
    const TileLayout L(N, M);
    // The lane terms of a phase (tile coordinates, LDS addresses) are recomputed at the phase's start from a thread id the
    // optimiser cannot see through: hoisted out of the horizon loop they would occupy well over a hundred registers,
    // which decides how many sub-problems share a CU.
#define WG_LANE_TERMS()                                                                               \
    int tid_p = tid;                                                                                  \
    asm volatile("" : "+v"(tid_p));                                                                   \
    const int wave = __builtin_amdgcn_readfirstlane(tid_p >> 6), lane = tid_p & 63, g = lane >> 4, c16 = lane & 15; \
    (void)wave; (void)g; (void)c16;

    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sP = lds + C::oP;          // [P|p] -> T1 -> [Q_xx|Q_x] -> V -> [P|p]
    double* sQ = sP;
    double* sG = lds + C::oG;          // T2 (at column MO) -> [Q_uu | Q_ux | Q_u]
    double* sK = lds + C::oK;          // [K|d]
    double* sT3 = lds + C::oT3;
    double* sMt = sG;                  // a2 scratch, after the S5 products (runs on into the [K|d] region)
    double* sAB = lds + C::oAB;        // [agent][row of the block][A_blk row | B_blk row], NSCP doubles each
    double* sTrash = lds + C::oEnd;

    const double mu = mu_arr[b];
    const double* base = FUSED ? nullptr : tiles + (int64_t)slot * (T + 1) * L.stride;
    int sing = 0;
    // FUSED: the step's plugin data (rows < m of the [K|d] region), the weights, this lane's share of (x, u) a step ahead
    double* sFx = sK + C::oFx;          // x
    double* sFe = sK + C::oFe;          // x - x_f
    double* sFu = sK + C::oFu;          // u
    double* sFg = sK + C::oFg;          // pair gradients [pair][3]
    double* sFh = sK + C::oFh;          // pair Hessians [pair][9]
    double* sFd = sK + C::oFd;          // per agent: sum of its pairs' Hessians [agent][9]
    double* sFs = sK + C::oFs;          // per agent: signed sum of its pairs' gradients [agent][3]
    double* sQQ = lds + C::oW;          // Q + Q^T [agent][NS][NS]
    double* sRR = sQQ + KA * NS * NS;   // R + R^T [agent][NC][NC]
    ItemParams IP{};
    const double* Xg = nullptr;
    const double* Ug = nullptr;
    double pxu = 0.0, f_xf = 0.0;
    const double wr = FUSED ? F.D.w_ref : 0.0, wp = FUSED ? F.D.w_prox : 0.0;
    if constexpr (FUSED) {
        IP = item_params(F.D, b);
        Xg = F.X + (int64_t)b * (T + 1) * N;
        Ug = F.U + (int64_t)b * T * M;
    }
    // this lane's role in the fused stage, once: agent lanes know their model, pair lanes their pair and its dimension count
    int f_model = 0, f_pi = 0, f_pj = 1, f_nd = 2;
    double f_radius = 0.0;
    if constexpr (FUSED) {
        if (tid < KA) f_model = IP.model[tid];
        if (tid >= 64 && tid - 64 < C::NPR) {
            int i = 0, rem = tid - 64;
            while (rem >= KA - 1 - i) { rem -= KA - 1 - i; ++i; }
            f_pi = i; f_pj = i + 1 + rem;
            f_nd = min(IP.n_dims[f_pi], IP.n_dims[f_pj]);  // cost.py:145
        }
        f_radius = IP.radius;
    }
    // FUSED, one step's plugin data (tiles.hpp phases 1a, 1b and the sums its phase 2 forms per entry) in two parts, neither
    // ending with a barrier: (x, x - x_f, u) from this lane's register into LDS -- done at the END of the previous step, next to
    // the store of P, so that it costs no barrier of its own -- and what is derived from them
    auto fused_put_xu = [&](bool terminal) {
        if (tid < N) { sFx[tid] = pxu; sFe[tid] = pxu - f_xf; }
        else if (tid < N + M) sFu[tid - N] = terminal ? 0.0 : pxu;
    };
    auto fused_derive = [&](bool terminal) {
        if (tid < KA) {
            if (!terminal) {
                wg_linearize_agent<NS, NC, NSCP>(f_model, sFx + tid * NS, sFu + tid * NC, F.D.dt, sAB + NS * tid * NSCP);
            }
        } else if (tid >= 64 && tid - 64 < C::NPR) {
            const int p = tid - 64;
            double g[3], H[9];
            pair_quadraticize(sFx + f_pi * NS, sFx + f_pj * NS, f_nd, f_radius, g, H);
#pragma unroll
            for (int c = 0; c < 3; ++c) sFg[p * 3 + c] = g[c];
#pragma unroll
            for (int c = 0; c < 9; ++c) sFh[p * 9 + c] = H[c];
        }
    };
    // ... and the per-agent sums of what the pair lanes left in LDS (behind a barrier after fused_derive).  Round 5: in the horizon
    // loop they are formed at the top of the S1 phase -- S1's products need neither, and the one consumer inside S1, l_x in the
    // p column's Q_x, moved into the S2 phase -- so the barrier that used to separate them from S1 is gone
    auto fused_sums = [&]() {
        if (KA > 1) {
            // per agent: the sums over its pairs, in combinations order (a serial chain of adds, as in the producer; the
            // loads are all issued first).  The lane's pair addresses and signs are formed HERE from a thread id the
            // optimiser cannot see through: as loop invariants of the horizon loop they were 2 k_a addresses + k_a masks per
            // lane, computed once and spilled (round 2: 30-50 spilled registers in every fused instantiation)
            int tid = tid_k;
            asm volatile("" : "+v"(tid));
            if (tid < 9 * KA) {
                const int a = tid / 9, c = tid - 9 * a;
                double h[KA];
#pragma unroll
                for (int o = 0; o < KA; ++o) {
                    const int oo = (o == a) ? (a == 0 ? 1 : 0) : o;
                    h[o] = sFh[((oo < a) ? pair_index(oo, a, KA) : pair_index(a, oo, KA)) * 9 + c];
                }
                double acc = 0.0;
#pragma unroll
                for (int o = 0; o < KA; ++o) acc = (o == a) ? acc : acc + h[o];
                sFd[tid] = acc;
            } else if (tid < 12 * KA) {
                const int e = tid - 9 * KA, a = e / 3, c = e - 3 * a;
                double gg[KA];
#pragma unroll
                for (int o = 0; o < KA; ++o) {
                    const int oo = (o == a) ? (a == 0 ? 1 : 0) : o;
                    const double v = sFg[((oo < a) ? pair_index(oo, a, KA) : pair_index(a, oo, KA)) * 3 + c];
                    gg[o] = (oo < a) ? -v : v;
                }
                double acc = 0.0;
#pragma unroll
                for (int o = 0; o < KA; ++o) acc = (o == a) ? acc : acc + gg[o];
                sFs[e] = acc;
            }
        }
    };

    for (int e = tid; e < C::total; e += kWgThreads) lds[e] = 0.0;
    wg_barrier();
    if constexpr (FUSED) {
        // the weights, once; then the terminal condition p = l_x(T), P = l_xx(T) (control.py:125-129) with Q_f
        for (int e = tid; e < KA * NS * NS; e += kWgThreads) {
            const int a = e / (NS * NS), r = e - a * NS * NS, li = r / NS, lj = r - li * NS;
            sQQ[e] = IP.Q[a * NS * NS + li * NS + lj] + IP.Q[a * NS * NS + lj * NS + li];
        }
        for (int e = tid; e < KA * NC * NC; e += kWgThreads) {
            const int a = e / (NC * NC), r = e - a * NC * NC, li = r / NC, lj = r - li * NC;
            sRR[e] = IP.R[a * NC * NC + li * NC + lj] + IP.R[a * NC * NC + lj * NC + li];
        }
        if (tid < N) { f_xf = IP.xf[tid]; pxu = Xg[(int64_t)T * N + tid]; }
        fused_put_xu(true);
        wg_barrier();
        fused_derive(true);
        wg_barrier();
        fused_sums();
        wg_barrier();
        for (int e = tid; e < N * N; e += kWgThreads) {
            const int i = e / N, j = e - i * N;
            const int ai = i / NS, aj = j / NS, li = i - ai * NS, lj = j - aj * NS;
            double v = 0.0;
            if (ai == aj) {
                const double* Mf = IP.Qf + ai * NS * NS;
                v = wr * (Mf[li * NS + lj] + Mf[lj * NS + li]);
            }
            if (KA > 1 && li < 3 && lj < 3) {
                double acc = 0.0;
                if (ai == aj) acc = sFd[ai * 9 + li * 3 + lj];
                else acc += -sFh[((ai < aj) ? pair_index(ai, aj, KA) : pair_index(aj, ai, KA)) * 9 + li * 3 + lj];
                v += wp * acc;
            }
            sP[i * LP + j] = v;
        }
        for (int j = tid; j < N; j += kWgThreads) {
            const int a = j / NS, lj = j - a * NS;
            const double* Mf = IP.Qf + a * NS * NS;
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) v += sFe[a * NS + i] * (Mf[i * NS + lj] + Mf[lj * NS + i]);
            v = wr * v;
            if (KA > 1 && lj < 3) v += wp * sFs[a * 3 + lj];
            sP[j * LP + N] = v;
        }
        if (tid < N) pxu = Xg[(int64_t)(T - 1) * N + tid];
        else if (tid < N + M) pxu = Ug[(int64_t)(T - 1) * M + (tid - N)];
        wg_barrier();           // the terminal step's x - x_f, sums have been read: step T - 1's (x, u) take their place
        fused_put_xu(false);
        if (T > 1) {
            if (tid < N) pxu = Xg[(int64_t)(T - 2) * N + tid];
            else if (tid < N + M) pxu = Ug[(int64_t)(T - 2) * M + (tid - N)];
        }
    } else {
        const double* rec = base + (int64_t)T * L.stride;
        for (int e = tid; e < N * N; e += kWgThreads) {
            const int i = e / N, j = e - i * N;
            sP[i * LP + j] = rec[L.oLxx + e];
        }
        for (int i = tid; i < N; i += kWgThreads) sP[i * LP + N] = rec[L.oLx + i];
    }

    // ---- S0 staging pattern of the diagonal blocks of [A|B]
    double* ab_dst[C::ABR];
    int ab_src[C::ABR];
#pragma unroll
    for (int q = 0; q < C::ABR; ++q) {
        const int e = tid + kWgThreads * q;
        const int ag = e / (NS * NSC), rem = e - ag * (NS * NSC), l = rem / NSC, c = rem - l * NSC;
        const int row = NS * ag + l;
        ab_dst[q] = (e < C::ABN) ? sAB + row * NSCP + c : sTrash + (tid & 1);
        ab_src[q] = (e < C::ABN) ? L.oA + row * L.ldAB + (c < NS ? NS * ag + c : N + NC * ag + (c - NS)) : 0;
    }
    double nAB[C::ABR];
    auto prefetch_ab = [&](int t) {
        const double* rec = base + (int64_t)t * L.stride;
#pragma unroll
        for (int q = 0; q < C::ABR; ++q) nAB[q] = rec[ab_src[q]];
    };
    // S1 work items: (agent, 2 columns of [P|p]); S2 work items: (agent, 2 rows of T)
#define WG_ITEMS_S1()                                                                                 \
    int ag1[C::R1R], j01[C::R1R];                                                                     \
    _Pragma("unroll") for (int r = 0; r < C::R1R; ++r) {                                              \
        const int w = min(tid_p + kWgThreads * r, C::NI1 - 1);                                        \
        ag1[r] = w / C::CG; j01[r] = 2 * (w - ag1[r] * C::CG);                                        \
    }
#define WG_ITEMS_S2()                                                                                 \
    int ag2[C::R2R], ip2[C::R2R][C::RPL];                                                             \
    _Pragma("unroll") for (int r = 0; r < C::R2R; ++r) {                                              \
        const int w = min(tid_p + kWgThreads * r, C::NI2 - 1);                                        \
        ag2[r] = w / C::RG;                                                                           \
        const int rg = w - ag2[r] * C::RG;                                                            \
        _Pragma("unroll") for (int q = 0; q < C::RPL; ++q) ip2[r][q] = min(C::RPL * rg + q, NM - 1);  \
    }
    // S3: lanes 0..M-1 hold Q_uu's columns, lanes M..63 this wavefront's right-hand sides
    constexpr int RW = 64 - M;
    constexpr int K_PAIRS = M * N / 2, K_ROUNDS = (K_PAIRS + kWgThreads - 1) / kWgThreads;

    // FUSED: where this lane's S2 work items find their l-values, once.  Row ip of agent ag's column block takes w_ref (Q + Q^T) or
    // w_ref (R + R^T) if the row is the agent's own, and -+ w_prox H of a pair (or + the agent's sum of them) on the position
    // entries.  Bits 0-9: the weights' row (doubles from sQQ), 10-21: the Hessian row (doubles from [K|d]), 24: own state row,
    // 25: own control row, 26: position row of a team, 27: own.  (Formed per step from the thread id -- two divisions, a pair
    // index and a dozen exec-mask regions per row -- this was more than half of the fused S2 phase's 1 375 instructions.)
    unsigned lvd[FUSED ? C::R2R : 1][C::RPL];
    if constexpr (FUSED) {
        static_assert(!FUSED || (C::szW < 1024 && M * C::LK < 4096), "the packed l-value descriptor's fields");
#pragma unroll
        for (int r = 0; r < C::R2R; ++r) {
            const int w = min(tid + kWgThreads * r, C::NI2 - 1);
            const int ag = w / C::RG, rg = w - ag * C::RG;
#pragma unroll
            for (int q = 0; q < C::RPL; ++q) {
                const int ip = min(C::RPL * rg + q, NM - 1);
                const bool xrow = ip < N;
                const int ai = xrow ? ip / NS : (ip - N) / NC, li = xrow ? ip - ai * NS : (ip - N) - ai * NC;
                const bool own = ai == ag;
                const int woff = xrow ? (ag * NS + li) * NS : KA * NS * NS + (ag * NC + li) * NC;
                const int pidx = own ? 0 : ((ai < ag) ? pair_index(ai, ag, KA) : pair_index(ag, ai, KA));
                const bool prox = KA > 1 && xrow && li < 3;
                const int hoff = prox ? (own ? C::oFd + ag * 9 : C::oFh + pidx * 9) + li * 3 : 0;
                unsigned dsc = (unsigned)woff | ((unsigned)hoff << 10) | ((xrow && own) ? 1u << 24 : 0u) | ((!xrow && own) ? 1u << 25 : 0u) |
                               (prox ? 1u << 26 : 0u) | (own ? 1u << 27 : 0u);
                asm volatile("" : "+v"(dsc));
                lvd[r][q] = dsc;
            }
        }
    }

    if constexpr (!FUSED) prefetch_ab(T - 1);
    wg_barrier();

#ifdef DPILQR_PHASE_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memtime();
#define WPHASE(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[i] += now_ - ph_t; ph_t = now_; }
#elif defined(DPILQR_PHASE_MARKS)   // assembly listings only: where each phase ends (hipcc -S)
#define WPHASE(i) asm volatile("; ==== end of phase " #i);
#else
#define WPHASE(i)
#endif
    for (int t = T - 1; t >= 0; --t) {
        const int tn = t > 0 ? t - 1 : 0;
        const double* rec = FUSED ? nullptr : base + (int64_t)t * L.stride;
        // ---- S0
        int tid_p = tid;
        asm volatile("" : "+v"(tid_p));
        WG_ITEMS_S1()
        double nX[C::R1R][NSC];
        if constexpr (FUSED) {
            fused_derive(false);    // (x, u) of this step were put in place at the end of the previous one
        } else {
#pragma unroll
        for (int q = 0; q < C::ABR; ++q) *ab_dst[q] = nAB[q];
        prefetch_ab(tn);
        // this step's [l_x | l_u]: requested now, used by the S1 epilogue
#pragma unroll
        for (int r = 0; r < C::R1R; ++r) {
#pragma unroll
            for (int c = 0; c < NSC; ++c) nX[r][c] = 0.0;
            if (j01[r] == N) {
                ld_row<NS, AL>(rec + L.oLx + NS * ag1[r], nX[r]);
                ld_row<NC, AL>(rec + L.oLu + NC * ag1[r], nX[r] + NS);
            }
        }
        }
        wg_barrier();
        WPHASE(0)

        __builtin_amdgcn_s_setprio(1);   // vector-pipe phases win arbitration over another workgroup's MFMA phases
        if constexpr (FUSED) fused_sums();
        // ---- S1: [A|B]^T [P|p], block diagonal.  T1 replaces P in place (this work item is the only reader and the only
        // writer of its agent's rows of its two columns); T2 goes where S2 will turn it into Q_ux
#pragma unroll
        for (int r = 0; r < C::R1R; ++r) {
            const int ag = ag1[r], j0 = j01[r];
            double acc[NSC * 2];   // [i][c]
            if constexpr (STRUCT4) {
                // The four-state family's blocks (DoubleIntDynamics4D, UnicycleDynamics4D; models.hpp jac): A = I + dt A_c with
                // A_c's entries (0,2), (1,2), (0,3), (1,3) free and nothing else, B = dt [0; 0; I].  Of the four-term chains of the
                // general form below only these terms are not a multiplication by an exact 0 or 1 -- dropped or written as what
                // they are, the results are the chains' bit for bit for FINITE P (a zero's sign apart): 8 instead of 48 multiply-adds
                // per item.  A non-finite P is another matter: the general chain turns 0 * Inf into NaN in every entry of the row,
                // the structured form only where P's non-finite entry really enters -- a diverged sweep's garbage spreads
                // differently (the item's status is the same: its costs are NaN either way and the line search rejects them).
                double pr4[4][2];
#pragma unroll
                for (int l = 0; l < 4; ++l) ld_row<2, true>(sP + (NS * ag + l) * LP + j0, pr4[l]);
                const double* ab0 = sAB + (NS * ag) * NSCP;
                double r0[2], r1[2];
                ld_row<2, true>(ab0 + 2, r0);              // a02, a03
                ld_row<2, true>(ab0 + NSCP + 2, r1);       // a12, a13
                const double b20 = ab0[2 * NSCP + 4], b31 = ab0[3 * NSCP + 5];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    acc[0 + c] = pr4[0][c];
                    acc[2 + c] = pr4[1][c];
                    acc[4 + c] = fma(r1[0], pr4[1][c], r0[0] * pr4[0][c]) + pr4[2][c];
                    acc[6 + c] = fma(r1[1], pr4[1][c], r0[1] * pr4[0][c]) + pr4[3][c];
                    acc[8 + c] = b20 * pr4[2][c];
                    acc[10 + c] = b31 * pr4[3][c];
                }
            } else {
            double ab[2][NSC], pr[2][2];   // the operand rows of l and l + 1
            ld_row<NSC, true>(sAB + (NS * ag) * NSCP, ab[0]);
            ld_row<2, true>(sP + (NS * ag) * LP + j0, pr[0]);
#pragma unroll
            for (int l = 0; l < NS; ++l) {
                if (l + 1 < NS) {
                    ld_row<NSC, true>(sAB + (NS * ag + l + 1) * NSCP, ab[(l + 1) & 1]);
                    ld_row<2, true>(sP + (NS * ag + l + 1) * LP + j0, pr[(l + 1) & 1]);
                }
#pragma unroll
                for (int i = 0; i < NSC; ++i)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[2 * i + c] = (l == 0) ? ab[l & 1][i] * pr[l & 1][c] : fma(ab[l & 1][i], pr[l & 1][c], acc[2 * i + c]);
                pin_regs(acc);
            }
            }
            // T2 rows: + mu B[j][c]   (quirk Q6: B^T (P + mu I) = B^T P + mu B^T); row j of B is zero outside agent j / NS
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int j = min(j0 + c, N - 1);
                double bm[NC];
                ld_row<NC, AL>(sAB + j * NSCP + NS, bm);
                const bool on = (j0 + c < N) && (j / NS == ag);
#pragma unroll
                for (int i = 0; i < NC; ++i) acc[2 * (NS + i) + c] = fma(mu, on ? bm[i] : 0.0, acc[2 * (NS + i) + c]);
            }
            if (tid_p + kWgThreads * r < C::NI1) {
                if (j0 < N) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) st_row<2, true>(sP + (NS * ag + i) * LP + j0, acc + 2 * i);
#pragma unroll
                    for (int i = 0; i < NC; ++i) st_row<2, true>(sG + (NC * ag + i) * LG + MO + j0, acc + 2 * (NS + i));
                } else {   // p column: Q_x = l_x + A^T p ; Q_u = l_u + B^T p
                    if constexpr (FUSED) {   // A^T p, B^T p now; l_x, l_u are added in the S2 phase (the sums they need are being formed)
#pragma unroll
                        for (int i = 0; i < NS; ++i) sP[(NS * ag + i) * LP + N] = acc[2 * i];
#pragma unroll
                        for (int i = 0; i < NC; ++i) sG[(NC * ag + i) * LG + MO + N] = acc[2 * (NS + i)];
                    } else {
#pragma unroll
                    for (int i = 0; i < NS; ++i) sP[(NS * ag + i) * LP + N] = nX[r][i] + acc[2 * i];
#pragma unroll
                    for (int i = 0; i < NC; ++i) sG[(NC * ag + i) * LG + MO + N] = nX[r][NS + i] + acc[2 * (NS + i)];
                    }
                }
            }
        }
        // the S2 l-values: requested here (not at the top of the step: they would be held across S1); FUSED: formed in S2
        double nL[C::R2R][C::RPL][NSC];
        if constexpr (!FUSED) {
        WG_ITEMS_S2()
#pragma unroll
        for (int r = 0; r < C::R2R; ++r)
#pragma unroll
            for (int q = 0; q < C::RPL; ++q) {
                const int ip = ip2[r][q];
                if (ip < N) {
                    ld_row<NS, ALA>(rec + L.oLxx + ip * N + NS * ag2[r], nL[r][q]);
#pragma unroll
                    for (int c = 0; c < NC; ++c) nL[r][q][NS + c] = 0.0;
                } else {
                    ld_row<NS, AL>(rec + L.oLux + (ip - N) * L.ldUG + NS * ag2[r], nL[r][q]);
                    ld_row<NC, AL>(rec + L.oLuu + (ip - N) * L.ldUG + NC * ag2[r], nL[r][q] + NS);
                }
            }
        }
        wg_barrier();
        WPHASE(1)

        // ---- S2: [T1;T2][A|B] + l-values -> Q_xx (rows < n), [Q_uu | Q_ux] (rows >= n); the T1 B block is dropped.
        // In place: a work item reads its rows' entries of its agent's column block and writes the same entries
        {
        int tid_p = tid;
        asm volatile("" : "+v"(tid_p));
        WG_ITEMS_S2()
#pragma unroll
        for (int r = 0; r < C::R2R; ++r) {
            const int ag = ag2[r];
            double acc[C::RPL * NSC], tv[C::RPL][NS];   // acc [q][c]
#pragma unroll
            for (int q = 0; q < C::RPL; ++q) {
                const int ip = ip2[r][q];
                ld_row<NS, ALA>((ip < N ? sP + ip * LP : sG + (ip - N) * LG + MO) + NS * ag, tv[q]);
            }
            if constexpr (FUSED) {
                // this work item's l-values, formed here (their LDS reads run under the reduction below):
                // L_xx = w_ref blockdiag(Q + Q^T) + w_prox sum_pairs(+-H) (cost.py:228-237, 160-169), L_ux = 0,
                // L_uu = w_ref blockdiag(R + R^T) -- its rows, its agent's column block
#pragma unroll
                for (int q = 0; q < C::RPL; ++q) {
                    const unsigned dsc = lvd[r][q];
                    const double* wrow = sQQ + (dsc & 1023u);
                    const double* hrow = sK + ((dsc >> 10) & 4095u);
                    const bool own_x = (dsc & (1u << 24)) != 0, own_u = (dsc & (1u << 25)) != 0, prox = (dsc & (1u << 26)) != 0,
                               own = (dsc & (1u << 27)) != 0;
                    double w[NS], h[3];
#pragma unroll
                    for (int c = 0; c < NS; ++c) w[c] = wrow[c];     // (a control row's NC entries and what follows them: inside LDS)
#pragma unroll
                    for (int c = 0; c < 3; ++c) h[c] = hrow[c];
#pragma unroll
                    for (int lj = 0; lj < NS; ++lj) {
                        double v = own_x ? wr * w[lj] : 0.0;
                        if (lj < 3) {
                            const double pa = own ? h[lj] : -h[lj];
                            const double vp = v + wp * pa;
                            v = prox ? vp : v;
                        }
                        nL[r][q][lj] = v;
                    }
#pragma unroll
                    for (int lj = 0; lj < NC; ++lj) nL[r][q][NS + lj] = own_u ? wr * w[lj] : 0.0;
                }
            }
            if constexpr (STRUCT4) {   // see S1: columns 0, 1 of T A are T's, 2 and 3 two free terms plus T's own, T B = T_2 b20, T_3 b31
                const double* ab0 = sAB + (NS * ag) * NSCP;
                double r0[2], r1[2];
                ld_row<2, true>(ab0 + 2, r0);
                ld_row<2, true>(ab0 + NSCP + 2, r1);
                const double b20 = ab0[2 * NSCP + 4], b31 = ab0[3 * NSCP + 5];
#pragma unroll
                for (int q = 0; q < C::RPL; ++q) {
                    acc[q * NSC + 0] = tv[q][0];
                    acc[q * NSC + 1] = tv[q][1];
                    acc[q * NSC + 2] = fma(tv[q][1], r1[0], tv[q][0] * r0[0]) + tv[q][2];
                    acc[q * NSC + 3] = fma(tv[q][1], r1[1], tv[q][0] * r0[1]) + tv[q][3];
                    acc[q * NSC + 4] = tv[q][2] * b20;
                    acc[q * NSC + 5] = tv[q][3] * b31;
                }
            } else {
            double ab[2][NSC];
            ld_row<NSC, true>(sAB + (NS * ag) * NSCP, ab[0]);
#pragma unroll
            for (int l = 0; l < NS; ++l) {
                if (l + 1 < NS) ld_row<NSC, true>(sAB + (NS * ag + l + 1) * NSCP, ab[(l + 1) & 1]);
#pragma unroll
                for (int q = 0; q < C::RPL; ++q)
#pragma unroll
                    for (int c = 0; c < NSC; ++c)
                        acc[q * NSC + c] = (l == 0) ? tv[q][l] * ab[l & 1][c] : fma(tv[q][l], ab[l & 1][c], acc[q * NSC + c]);
                pin_regs(acc);
            }
            }
            if (tid_p + kWgThreads * r < C::NI2) {
#pragma unroll
                for (int q = 0; q < C::RPL; ++q) {
                    const int ip = ip2[r][q];
                    if (q > 0 && ip == ip2[r][q - 1]) continue;   // the clamped last row of an odd n + m
                    double o[NSC];
#pragma unroll
                    for (int c = 0; c < NSC; ++c) o[c] = nL[r][q][c] + acc[q * NSC + c];
                    if (ip < N) {
                        st_row<NS, ALA>(sQ + ip * LQ + NS * ag, o);
                    } else {
                        st_row<NS, ALA>(sG + (ip - N) * LG + MO + NS * ag, o);
                        st_row<NC, AL>(sG + (ip - N) * LG + NC * ag, o + NS);
                    }
                }
            }
        }
        }
        if constexpr (FUSED) {
            // Q_x = l_x + A^T p, Q_u = l_u + B^T p: the l-part, one entry per lane (S2's work items never touch column n).
            // l_x = w_ref e^T (Q + Q^T) + w_prox sum_pairs(+-g) ; l_u = w_ref u^T (R + R^T) -- the expressions and orders S1 had
            int tid_q = tid;
            asm volatile("" : "+v"(tid_q));
            if (tid_q < N) {
                const int ag = tid_q / NS, lj = tid_q - ag * NS;
                double v = 0.0;
#pragma unroll
                for (int i = 0; i < NS; ++i) v += sFe[ag * NS + i] * sQQ[(ag * NS + i) * NS + lj];
                v = wr * v;
                if (KA > 1 && lj < 3) v += wp * sFs[ag * 3 + lj];
                sP[tid_q * LP + N] = v + sP[tid_q * LP + N];
            } else if (tid_q < N + M) {
                const int a = tid_q - N, ag = a / NC, lj = a - ag * NC;
                double v = 0.0;
#pragma unroll
                for (int i = 0; i < NC; ++i) v += sFu[ag * NC + i] * sRR[(ag * NC + i) * NC + lj];
                sG[a * LG + MO + N] = wr * v + sG[a * LG + MO + N];
            }
        }
        // [K | d]'s reduction-padding rows must read as zero (the previous step's a2 may have reached them)
        for (int e = tid; e < (C::KROWS - M) * LK; e += kWgThreads) sK[M * LK + e] = 0.0;
        wg_barrier();
        WPHASE(2)

        // ---- S3: [K | d] = -Q_uu^-1 [Q_ux | Q_u] : LU with partial pivoting in registers, once per wavefront
        // By blocks first (every wavefront that owns a 16-column tile of right-hand sides); if that declines (a diagonal
        // pivot too small for its column), LAPACK-order partial pivoting in registers, for which only as many wavefronts as
        // the right-hand sides need take part (one up to m = 21, two beyond): further copies of the factorisation would
        // only take issue slots from the sub-problems that share the CU.
        constexpr int NWLU = (NP + RW - 1) / RW;
        int lu_wave = tid >> 6;
        asm volatile("" : "+v"(lu_wave));
        lu_wave = __builtin_amdgcn_readfirstlane(lu_wave);
        bool lu_needed = lu_wave < NWLU;
        if constexpr (C::GJ) {
            static_assert(!C::GJ || NWLU <= T_NP, "the wavefronts of the fall-back must have seen the blocked attempt fail");
            if (lu_wave < T_NP) {
                WG_LANE_TERMS()
                const bool declined = gj_blocked<M, MO, LG, LK, NP>(sG, sK, lds + (C::PAN_IN_AB ? C::oAB : C::oPan) + 128 * wave, wave, lane);
                lu_needed = lu_needed && declined;
#if defined(DPILQR_PHASE_STAMPS) && !defined(DPILQR_S3_SPLIT)
                if (!declined) ph[6] += 1000;   // diagnostic: share of the steps solved by blocks (per mille)
#endif
            }
        }
#if defined(DPILQR_PHASE_STAMPS) && defined(DPILQR_S3_SPLIT)   // slots 6, 7: the blocked elimination, the fall-back (instead of the two shares)
        WPHASE(6)
#endif
        if (lu_needed) {
            WG_LANE_TERMS()
            // out of line: the register LU holds m columns + its broadcasts (250 registers, hundreds of scalar temporaries at
            // m = 30) and, where the blocked elimination serves the size, runs on a few per cent of the steps only -- inlined
            // it set the register allocation of the whole kernel (round 2: every fused instantiation spilled)
            const int flags = lu_fallback_wg<M, N, NP, MO, LG, LK, RW>(sG, sK, wave, lane);
            if (flags & 1) sing = 1;
#if defined(DPILQR_PHASE_STAMPS) && !defined(DPILQR_S3_SPLIT)
            if (!C::GJ && (flags & 2)) ph[6] += 1000;   // diagnostic: share of the steps with the search-free elimination
            if (flags & 4) ph[7] += 1000;               // ... and of the steps in which partial pivoting moved a row
#endif
        }
#if defined(DPILQR_PHASE_STAMPS) && defined(DPILQR_S3_SPLIT)
        WPHASE(7)
#endif
        wg_barrier();
        {
            int tid_p = tid;
            asm volatile("" : "+v"(tid_p));
            double* Kt = Kout + (gslot * T + t) * M * N;
            double* dt_ = dout + (gslot * T + t) * M;
#pragma unroll
            for (int q = 0; q < K_ROUNDS; ++q) {
                const int e = 2 * min(tid_p + kWgThreads * q, K_PAIRS - 1);
                store_v2d_nt(Kt + e, *reinterpret_cast<const v2d*>(sK + (e / N) * LK + (e % N)));
            }
            const int di = min(tid_p, M - 1);
            store_f64_nt(dt_ + di, sK[di * LK + N]);
        }
        WPHASE(3)

        __builtin_amdgcn_s_setprio(0);
        // ---- S4: T3^T[c][i] = sum_a Q_uu[a][c] K[a][i]
        // ---- S5: a1 = T3 [K|d] ; a2 = [K|d]^T [Q_ux|Q_u] ; V = ((Q + a1) + a2) + a2^T   (rows < n, cols <= n)
        v4d a1[C::TPW5], a2[C::TPW5];
        if constexpr (C::T3REG) {
            WG_LANE_TERMS()
            static_assert(!C::T3REG || (C::TPW5 == T_NP && C::T_M <= 2), "wavefront w owns row tile w");
            v4d t3[C::T_M];
#pragma unroll
            for (int it = 0; it < C::T_M; ++it) {
                const double* px = sG + g * LG + c16 + 16 * it;
                const double* py = sK + g * LK + c16 + 16 * wave;
                t3[it] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < MK / 4; ++ks) t3[it] = mfma_f64(px[ks * 4 * LG], py[ks * 4 * LK], t3[it]);
            }
            WPHASE(4)
#pragma unroll
            for (int q = 0; q < C::TPW5; ++q) {   // tile (row tile wave, column tile q)
                const double* pkj = sK + g * LK + c16 + 16 * q;
                const double* pki = sK + g * LK + c16 + 16 * wave;
                const double* pgj = sG + g * LG + MO + c16 + 16 * q;
                a1[q] = v4d{0.0, 0.0, 0.0, 0.0};
                a2[q] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < MK / 4; ++ks) {
                    a1[q] = mfma_f64(t3[ks / 4][ks % 4], pkj[ks * 4 * LK], a1[q]);
                    a2[q] = mfma_f64(pki[ks * 4 * LK], pgj[ks * 4 * LG], a2[q]);
                }
            }
        } else {
        {
            WG_LANE_TERMS()
            v4d acc[C::TPW4];
#pragma unroll
            for (int q = 0; q < C::TPW4; ++q) {
                const int tl = min(wave + 4 * q, C::NT4 - 1);
                const int it = tl / T_N, jt = tl - it * T_N;
                const double* px = sG + g * LG + c16 + 16 * it;
                const double* py = sK + g * LK + c16 + 16 * jt;
                acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < MK / 4; ++ks) acc[q] = mfma_f64(px[ks * 4 * LG], py[ks * 4 * LK], acc[q]);
            }
#pragma unroll
            for (int q = 0; q < C::TPW4; ++q) {
                const int tl = min(wave + 4 * q, C::NT4 - 1);
                const int it = tl / T_N, jt = tl - it * T_N;
                const int j = 16 * jt + c16;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 16 * it + g + 4 * v;
                    if (i < M && j < N) sT3[i * N + j] = acc[q][v];
                }
            }
        }
        wg_barrier();
        WPHASE(4)
        {
            WG_LANE_TERMS()
#pragma unroll
            for (int q = 0; q < C::TPW5; ++q) {
                const int tl = min(wave + 4 * q, C::NT5 - 1);
                const int it = tl / T_NP, jt = tl - it * T_NP;
                const double* pt3 = sT3 + g * N + c16 + 16 * it;
                const double* pkj = sK + g * LK + c16 + 16 * jt;
                const double* pki = sK + g * LK + c16 + 16 * it;
                const double* pgj = sG + g * LG + MO + c16 + 16 * jt;
                a1[q] = v4d{0.0, 0.0, 0.0, 0.0};
                a2[q] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < MK / 4; ++ks) {
                    a1[q] = mfma_f64(pt3[ks * 4 * N], pkj[ks * 4 * LK], a1[q]);
                    a2[q] = mfma_f64(pki[ks * 4 * LK], pgj[ks * 4 * LG], a2[q]);
                }
            }
        }
        }
        {
            WG_LANE_TERMS()
            wg_barrier();   // every operand read of G, [K|d], T3^T is done: a2 goes there for the transposed read
            // No exec-mask regions from here to the store of P: a lane whose element lies outside the matrix loads from
            // offset 0 and stores to its own word of the trash region, so every pass is loads | arithmetic | stores with
            // one wait each.  (Written with `if (i < n && j < n)` around each element this epilogue was 144 branches and 55
            // full LDS waits per step -- a quarter of the step's instructions.)
            double* const trash = sTrash + lane;
            double W[C::TPW5][4], ld0[C::TPW5][4], ld1[C::TPW5][4];
            // the loads of two tiles at a time where the registers allow it (two sub-problems per CU), tile by tile otherwise
            constexpr int CH6 = C::OCC >= 3 ? 1 : (C::TPW5 % 2 == 0 ? 2 : C::TPW5);
#pragma unroll
            for (int q0 = 0; q0 < C::TPW5; q0 += CH6) {
#pragma unroll
                for (int q = q0; q < q0 + CH6; ++q) {
                    const int tl = C::T3REG ? wave * T_NP + q : wave + 4 * q;
                    const int tc = min(tl, C::NT5 - 1);
                    const int it = tc / T_NP, jt = tc - it * T_NP;
                    const int j = 16 * jt + c16;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int i = 16 * it + g + 4 * v;
                        const bool in_w = tl < C::NT5 && i < N && j <= N;
                        ld0[q][v] = sQ[in_w ? i * LQ + j : 0];
                    }
                }
#pragma unroll
                for (int q = q0; q < q0 + CH6; ++q) {
                    const int tl = C::T3REG ? wave * T_NP + q : wave + 4 * q;
                    const int tc = min(tl, C::NT5 - 1);
                    const int it = tc / T_NP, jt = tc - it * T_NP;
                    const int j = 16 * jt + c16;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int i = 16 * it + g + 4 * v;
                        const bool live = tl < C::NT5;
                        W[q][v] = (ld0[q][v] + a1[q][v]) + a2[q][v];
                        *((live && i < NP && j < N) ? sMt + i * LM + j : trash) = a2[q][v];
                        *((live && i < N && j < N) ? sQ + i * LQ + j : trash) = W[q][v];
                    }
                }
            }
            wg_barrier();
            // ---- S6: V = W + a2^T ; V^T from the mirrored pair (same operands, same order) ; P <- (V + V^T)/2.
            // P takes W's place: every transposed read of W happens before the barrier, every store after it.
#pragma unroll
            for (int q0 = 0; q0 < C::TPW5; q0 += CH6) {
#pragma unroll
                for (int q = q0; q < q0 + CH6; ++q) {
                    const int tl = C::T3REG ? wave * T_NP + q : wave + 4 * q;
                    const int tc = min(tl, C::NT5 - 1);
                    const int it = tc / T_NP, jt = tc - it * T_NP;
                    const int j = 16 * jt + c16;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int i = 16 * it + g + 4 * v;
                        const bool in_w = tl < C::NT5 && i < N && j <= N;
                        ld0[q][v] = sMt[in_w ? j * LM + i : 0];
                        ld1[q][v] = sQ[(in_w && j < N) ? j * LQ + i : 0];
                    }
                }
#pragma unroll
                for (int q = q0; q < q0 + CH6; ++q) {
                    const int tl = C::T3REG ? wave * T_NP + q : wave + 4 * q;
                    const int tc = min(tl, C::NT5 - 1);
                    const int jt = tc - (tc / T_NP) * T_NP;
                    const int j = 16 * jt + c16;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const double V = W[q][v] + ld0[q][v];
                        const double Vt = ld1[q][v] + a2[q][v];
                        W[q][v] = j < N ? 0.5 * (V + Vt) : V;
                    }
                }
                if constexpr (CH6 < C::TPW5) asm volatile("" ::: "memory");   // the next chunk's loads stay behind this chunk's sums
            }
            wg_barrier();
#pragma unroll
            for (int q = 0; q < C::TPW5; ++q) {
                const int tl = C::T3REG ? wave * T_NP + q : wave + 4 * q;
                const int tc = min(tl, C::NT5 - 1);
                const int it = tc / T_NP, jt = tc - it * T_NP;
                const int j = 16 * jt + c16;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 16 * it + g + 4 * v;
                    *((tl < C::NT5 && i < N && j <= N) ? sP + i * LP + j : trash) = W[q][v];
                }
            }
        }
        if constexpr (FUSED) {
            if (t > 0) {            // the next step's (x, u): a2 has been read, rows < m of the [K|d] region are free
                fused_put_xu(false);
                if (t > 1) {
                    if (tid < N) pxu = Xg[(int64_t)(t - 2) * N + tid];
                    else if (tid < N + M) pxu = Ug[(int64_t)(t - 2) * M + (tid - N)];
                }
            }
        }
        wg_barrier();
        WPHASE(5)
    }
    if (singular && sing && tid == 0) singular[b] = 1;
#ifdef DPILQR_PHASE_STAMPS
    if (g_stamp_buf && tid == 0)
        for (int i = 0; i < 8; ++i) g_stamp_buf[4 * B + 8 * slot + i] = ph[i];
#endif
}

}  // namespace dpilqr
