// riccati_wg.hpp -- K2 for the larger clusters (n_x 24 .. 60): the Riccati backward sweep with ONE WORKGROUP of four
// wavefronts per sub-problem (ilqrSolver._backward_pass, control.py:116-148).
//
// riccati_mfma.hpp gives every sub-problem one wavefront and 20 KB of LDS; that stops at n_x = 20.  A 15-agent
// unicycle cluster (cfg3) has n_x = 60, n_u = 30: [A|B], [P|p], T, [Q_xx|Q_x], [Q_uu|Q_ux|Q_u] need 147 KB, a whole
// CU's LDS, and the work of one step (a 30 x 30 pivoted LU with 61 right-hand sides, 0.27 M dense FMAs) is enough
// for four wavefronts.  Same recursion, same association order as the reference (see riccati.hpp for the equations):
//
//   S0  [A|B](t) registers -> LDS (requested during the previous step); request this step's l-values
//   S1  [T1;T2 | A^T p;B^T p] = [A|B]^T [P|p]  (+ mu B^T on the B rows): block-diagonal [A|B] (the library's own
//       tiles, see riccati_mfma.hpp), NS terms per output, work items (agent, 2 columns) dealt to the 256 lanes
//   S2  [T1;T2][A|B] + l-values -> Q_xx, Q_ux, Q_uu: work items (agent, 2 rows)
//   S3  LU with partial pivoting, in registers: EVERY wavefront factorises Q_uu (lanes 0..m-1 hold its columns) and
//       carries its own 64-m of the n+1 right-hand sides, so no pivot or multiplier ever crosses a wavefront
//   S4  T3^T = Q_uu-contracted K            fp64 MFMA tiles dealt round-robin to the wavefronts
//   S5  a1 = T3 [K|d], a2 = [K|d]^T [Q_ux|Q_u], V = ((Q + a1) + a2) + a2^T          likewise
//   S6  P <- (V + V^T)/2
// LDS regions: [A|B] | [Q_xx|Q_x] | R2 = [P|p] -> T -> [K|d] + T3^T -> [P|p] | G = [Q_uu|Q_ux|Q_u] -> a2.  Phases are
// separated by workgroup barriers (s_barrier behind an LDS-only wait: the global prefetches stay in flight).
#pragma once
#include <hip/hip_runtime.h>

#include "riccati_mfma.hpp"

namespace dpilqr {

constexpr int kWgThreads = 256;

template <int N, int M, int NS, int NC>
struct WgCfg {
    static constexpr int NM = N + M, NP = N + 1, MK = round_up(M, 4), KA = N / NS, NSC = NS + NC;
    static constexpr int LAB = round_up(NM, 2), LP = round_up(NP, 2), LQ = LP, LG = round_up(M + NP, 2), LK = LP;
    static constexpr int LTB = LP, LM = N, KROWS = MK + 2;
    static constexpr int T_NP = (NP + 15) / 16, T_N = (N + 15) / 16, T_M = (M + 15) / 16;
    static constexpr int szAB = N * LAB, szQ = N * LQ;
    static constexpr int szKT3 = KROWS * LK + MK * N + 16;
    static constexpr int szR2a = NM * LTB > szKT3 ? NM * LTB : szKT3;
    static constexpr int szR2 = round_up(szR2a > N * LP ? szR2a : N * LP, 2);
    static constexpr int szG0 = MK * LG + round_up(N, 2);
    static constexpr int szG = round_up(szG0 > NP * LM ? szG0 : NP * LM, 2);
    static constexpr int oAB = 0, oQ = oAB + szAB, oR2 = oQ + szQ, oK = oR2, oT3 = oK + KROWS * LK, oP = oR2, oT = oR2;
    static constexpr int oG = oR2 + szR2, oQx = oG + MK * LG, oEnd = oG + szG;
    static constexpr int total = round_up(oEnd + 64, 2);   // + store target of idle lanes, wrapped tile reads
    static constexpr bool AL = (NS % 2 == 0) && (NC % 2 == 0);   // every block offset even: 16-byte vector accesses
    static constexpr bool ALA = (NS % 2 == 0);                   // ... at least the A-column blocks (offsets NS*agent)
    static constexpr int CG = (NP + 1) / 2;                  // S1: column groups of 2 over [P|p]
    static constexpr int NI1 = KA * CG, R1R = (NI1 + kWgThreads - 1) / kWgThreads;
    static constexpr int RPL = 2, RG = (NM + RPL - 1) / RPL;  // S2: row groups
    static constexpr int NI2 = KA * RG, R2R = (NI2 + kWgThreads - 1) / kWgThreads;
    static constexpr bool ABV = (NM % 2 == 0);                                 // rows of [A|B] split into 16-byte pairs
    static constexpr int ABE = ABV ? 2 : 1;                                    // doubles per staged element
    static constexpr int ABR = (N * NM / ABE + kWgThreads - 1) / kWgThreads;   // elements of [A|B] a lane stages
    static constexpr int NT4 = T_M * T_N, TPW4 = (NT4 + 3) / 4;              // S4 tiles, per wavefront
    static constexpr int NT5 = T_NP * T_NP, TPW5 = (NT5 + 3) / 4;            // S5 tiles, per wavefront
    static constexpr bool supported = (N % NS == 0) && (M == KA * NC) && (N % 2 == 0) && (M <= 32) && (64 - M > 0) &&
                                      (4 * (64 - M) >= NP) && (total * 8 <= 160 * 1024);
};

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int LEN, bool VEC>
__device__ __forceinline__ void ld_row(const double* p, double* out) {   // LEN doubles; VEC: p is 16-byte aligned, LEN even
    if constexpr (VEC) {
#pragma unroll
        for (int q = 0; q < LEN / 2; ++q) {
            const v2d v = *reinterpret_cast<const v2d*>(p + 2 * q);
            out[2 * q] = v.x; out[2 * q + 1] = v.y;
        }
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
    } else {
#pragma unroll
        for (int q = 0; q < LEN; ++q) p[q] = v[q];
    }
}

template <int N, int M, int NS, int NC>
__global__ __launch_bounds__(kWgThreads) void k_riccati_wg(
    int B, int T, const double* __restrict__ tiles, const double* __restrict__ mu_arr, double* __restrict__ Kout,
    double* __restrict__ dout, int32_t* __restrict__ singular, const int32_t* __restrict__ items,
    const int32_t* __restrict__ n_items, int gains_by_item) {
    using C = WgCfg<N, M, NS, NC>;
    constexpr int NM = C::NM, NP = C::NP, MK = C::MK, KA = C::KA, NSC = C::NSC, LAB = C::LAB, LP = C::LP, LQ = C::LQ;
    constexpr int LG = C::LG, LK = C::LK, LTB = C::LTB, LM = C::LM, T_NP = C::T_NP, T_N = C::T_N;
    constexpr bool AL = C::AL, ALA = C::ALA;
    const int slot = blockIdx.x;
    if (slot >= (n_items ? *n_items : B)) return;
    const int b = items ? items[slot] : slot;
    if (b >= B) return;
    const int64_t gslot = gains_by_item ? b : slot;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
    const TileLayout L(N, M);

    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sAB = lds + C::oAB;
    double* sQ = lds + C::oQ;
    double* sP = lds + C::oP;
    double* sT = lds + C::oT;
    double* sK = lds + C::oK;
    double* sT3 = lds + C::oT3;
    double* sG = lds + C::oG;
    double* sQx = lds + C::oQx;
    double* sMt = sG;                  // a2 scratch, after the S5 products
    double* sTrash = lds + C::oEnd;

    const double mu = mu_arr[b];
    const double* base = tiles + (int64_t)slot * (T + 1) * L.stride;
    int sing = 0;

    for (int e = tid; e < C::total; e += kWgThreads) lds[e] = 0.0;
    wg_barrier();
    {
        const double* rec = base + (int64_t)T * L.stride;
        for (int e = tid; e < N * N; e += kWgThreads) {
            const int i = e / N, j = e - i * N;
            sP[i * LP + j] = rec[L.oLxx + e];
        }
        for (int i = tid; i < N; i += kWgThreads) sP[i * LP + N] = rec[L.oLx + i];
    }

    // ---- S0 staging pattern of [A|B]
    double* ab_dst[C::ABR];
    int ab_src[C::ABR];
#pragma unroll
    for (int q = 0; q < C::ABR; ++q) {
        const int e = C::ABE * (tid + kWgThreads * q);
        const int row = e / NM, col = e - row * NM;
        ab_dst[q] = (e < N * NM) ? sAB + row * LAB + col : sTrash;
        ab_src[q] = (e < N * NM) ? L.oA + e : 0;
    }
    double nAB[C::ABR][C::ABE];
    auto prefetch_ab = [&](int t) {
        const double* rec = base + (int64_t)t * L.stride;
#pragma unroll
        for (int q = 0; q < C::ABR; ++q) ld_row<C::ABE, C::ABV>(rec + ab_src[q], nAB[q]);
    };
    // ---- S1 work items: (agent, 2 columns of [P|p]); S2 work items: (agent, 2 rows of T)
    int ag1[C::R1R], j01[C::R1R];
#pragma unroll
    for (int r = 0; r < C::R1R; ++r) {
        const int w = min(tid + kWgThreads * r, C::NI1 - 1);
        ag1[r] = w / C::CG; j01[r] = 2 * (w - ag1[r] * C::CG);
    }
    int ag2[C::R2R], ip2[C::R2R][C::RPL];
#pragma unroll
    for (int r = 0; r < C::R2R; ++r) {
        const int w = min(tid + kWgThreads * r, C::NI2 - 1);
        ag2[r] = w / C::RG;
        const int rg = w - ag2[r] * C::RG;
#pragma unroll
        for (int q = 0; q < C::RPL; ++q) ip2[r][q] = min(C::RPL * rg + q, NM - 1);
    }
    // ---- S3: lanes 0..M-1 hold Q_uu's columns, lanes M..63 this wavefront's right-hand sides
    constexpr int RW = 64 - M;
    const int s3_q = RW * wave + (lane - M);
    const bool s3_rhs = lane >= M && s3_q < NP;
    const int s3_col = lane < M ? lane : M + min(max(s3_q, 0), N);
    constexpr int K_PAIRS = M * N / 2, K_ROUNDS = (K_PAIRS + kWgThreads - 1) / kWgThreads;

    prefetch_ab(T - 1);
    wg_barrier();

    for (int t = T - 1; t >= 0; --t) {
        const int tn = t > 0 ? t - 1 : 0;
        const double* rec = base + (int64_t)t * L.stride;
        // ---- S0
#pragma unroll
        for (int q = 0; q < C::ABR; ++q) st_row<C::ABE, C::ABV>(ab_dst[q], nAB[q]);
        // this step's l-values: requested now, used by the S1 / S2 epilogues
        double nX[C::R1R][NSC];
#pragma unroll
        for (int r = 0; r < C::R1R; ++r) {
#pragma unroll
            for (int c = 0; c < NSC; ++c) nX[r][c] = 0.0;
            if (j01[r] == N) {
                ld_row<NS, AL>(rec + L.oLx + NS * ag1[r], nX[r]);
                ld_row<NC, AL>(rec + L.oLu + NC * ag1[r], nX[r] + NS);
            }
        }
        double nL[C::R2R][C::RPL][NSC];
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
        wg_barrier();

        __builtin_amdgcn_s_setprio(1);   // vector-pipe phases win arbitration over another workgroup's MFMA phases
        // ---- S1: [A|B]^T [P|p], block diagonal
        {
            double acc[C::R1R][NSC][2];
#pragma unroll
            for (int r = 0; r < C::R1R; ++r) {
                const int ag = ag1[r], j0 = j01[r];
#pragma unroll
                for (int l = 0; l < NS; ++l) {
                    double ab[NSC], pr[2];
                    ld_row<NS, ALA>(sAB + (NS * ag + l) * LAB + NS * ag, ab);
                    ld_row<NC, AL>(sAB + (NS * ag + l) * LAB + N + NC * ag, ab + NS);
                    ld_row<2, true>(sP + (NS * ag + l) * LP + j0, pr);
#pragma unroll
                    for (int i = 0; i < NSC; ++i)
#pragma unroll
                        for (int c = 0; c < 2; ++c) acc[r][i][c] = (l == 0) ? ab[i] * pr[c] : fma(ab[i], pr[c], acc[r][i][c]);
                }
                // T2 rows: + mu B[j][c]   (quirk Q6: B^T (P + mu I) = B^T P + mu B^T)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    double bm[NC];
                    ld_row<NC, AL>(sAB + min(j0 + c, N - 1) * LAB + N + NC * ag, bm);
#pragma unroll
                    for (int i = 0; i < NC; ++i) acc[r][NS + i][c] = fma(mu, (j0 + c < N) ? bm[i] : 0.0, acc[r][NS + i][c]);
                }
            }
            wg_barrier();   // every read of [P|p] is done: T takes its place
#pragma unroll
            for (int r = 0; r < C::R1R; ++r) {
                const int ag = ag1[r], j0 = j01[r];
#pragma unroll
                for (int i = 0; i < NS; ++i) st_row<2, true>(sT + (NS * ag + i) * LTB + j0, acc[r][i]);
#pragma unroll
                for (int i = 0; i < NC; ++i) st_row<2, true>(sT + (N + NC * ag + i) * LTB + j0, acc[r][NS + i]);
                if (j0 == N) {   // p column: Q_x = l_x + A^T p ; Q_u = l_u + B^T p
#pragma unroll
                    for (int i = 0; i < NS; ++i) sQx[NS * ag + i] = nX[r][i] + acc[r][i][0];
#pragma unroll
                    for (int i = 0; i < NC; ++i) sG[(NC * ag + i) * LG + M + N] = nX[r][NS + i] + acc[r][NS + i][0];
                }
            }
        }
        wg_barrier();

        // ---- S2: [T1;T2][A|B] + l-values -> Q_xx (rows < n), [Q_uu | Q_ux] (rows >= n); the T1 B block is dropped
#pragma unroll
        for (int r = 0; r < C::R2R; ++r) {
            const int ag = ag2[r];
            double acc[C::RPL][NSC], tv[C::RPL][NS];
#pragma unroll
            for (int q = 0; q < C::RPL; ++q) ld_row<NS, ALA>(sT + ip2[r][q] * LTB + NS * ag, tv[q]);
#pragma unroll
            for (int l = 0; l < NS; ++l) {
                double ab[NSC];
                ld_row<NS, ALA>(sAB + (NS * ag + l) * LAB + NS * ag, ab);
                ld_row<NC, AL>(sAB + (NS * ag + l) * LAB + N + NC * ag, ab + NS);
#pragma unroll
                for (int q = 0; q < C::RPL; ++q)
#pragma unroll
                    for (int c = 0; c < NSC; ++c) acc[q][c] = (l == 0) ? tv[q][l] * ab[c] : fma(tv[q][l], ab[c], acc[q][c]);
            }
#pragma unroll
            for (int q = 0; q < C::RPL; ++q) {
                const int ip = ip2[r][q];
                double o[NSC];
#pragma unroll
                for (int c = 0; c < NSC; ++c) o[c] = nL[r][q][c] + acc[q][c];
                if (ip < N) {
                    st_row<NS, ALA>(sQ + ip * LQ + NS * ag, o);
                } else {
                    st_row<NS, AL>(sG + (ip - N) * LG + M + NS * ag, o);
                    st_row<NC, AL>(sG + (ip - N) * LG + NC * ag, o + NS);
                }
            }
        }
        wg_barrier();
        // T is dead: [K | d] takes its place; its reduction-padding rows must read as zero.  Q_x joins Q_xx.
        for (int e = tid; e < (C::KROWS - M) * LK; e += kWgThreads) sK[M * LK + e] = 0.0;
        if (tid < N) sQ[tid * LQ + N] = sQx[tid];

        // ---- S3: [K | d] = -Q_uu^-1 [Q_ux | Q_u] : LU with partial pivoting in registers, once per wavefront
        {
            double v[M], invd[M];
#pragma unroll
            for (int r = 0; r < M; ++r) v[r] = sG[r * LG + s3_col];
            // strictly column-dominant Q_uu: no row moves, the elimination runs without the pivot search
            // (see S3 of k_riccati_mfma for the argument and the margin)
            double colsum = 0.0;
#pragma unroll
            for (int r = 0; r < M; ++r) colsum = colsum + fabs(v[r]);
            const double dg = fabs(sG[min(lane, M - 1) * (LG + 1)]);
            const bool dominant = lane >= M || dg * (1.0 - 0x1p-20) > colsum - dg;
            if (__builtin_amdgcn_ballot_w64(!dominant) == 0ull) lu_eliminate<false, M, false>(v, invd, sing);
            else lu_eliminate<true, M, false>(v, invd, sing);
#pragma unroll
            for (int r = M - 1; r >= 0; --r) {
                double s = v[r];
#pragma unroll
                for (int c = r + 1; c < M; ++c) s = fma(-readlane_f64(v[r], c), v[c], s);
                v[r] = s * invd[r];
            }
            if (s3_rhs) {
#pragma unroll
                for (int a = 0; a < M; ++a) sK[a * LK + s3_q] = -v[a];
            }
        }
        wg_barrier();
        {
            double* Kt = Kout + (gslot * T + t) * M * N;
            double* dt_ = dout + (gslot * T + t) * M;
#pragma unroll
            for (int q = 0; q < K_ROUNDS; ++q) {
                const int e = 2 * min(tid + kWgThreads * q, K_PAIRS - 1);
                store_v2d_nt(Kt + e, *reinterpret_cast<const v2d*>(sK + (e / N) * LK + (e % N)));
            }
            const int di = min(tid, M - 1);
            store_f64_nt(dt_ + di, sK[di * LK + N]);
        }
        prefetch_ab(tn);

        __builtin_amdgcn_s_setprio(0);
        // ---- S4: T3^T[c][i] = sum_a Q_uu[a][c] K[a][i]
        {
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

        // ---- S5: a1 = T3 [K|d] ; a2 = [K|d]^T [Q_ux|Q_u] ; V = ((Q + a1) + a2) + a2^T   (rows < n, cols <= n)
        {
            v4d a1[C::TPW5], a2[C::TPW5];
#pragma unroll
            for (int q = 0; q < C::TPW5; ++q) {
                const int tl = min(wave + 4 * q, C::NT5 - 1);
                const int it = tl / T_NP, jt = tl - it * T_NP;
                const double* pt3 = sT3 + g * N + c16 + 16 * it;
                const double* pkj = sK + g * LK + c16 + 16 * jt;
                const double* pki = sK + g * LK + c16 + 16 * it;
                const double* pgj = sG + g * LG + M + c16 + 16 * jt;
                a1[q] = v4d{0.0, 0.0, 0.0, 0.0};
                a2[q] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < MK / 4; ++ks) {
                    a1[q] = mfma_f64(pt3[ks * 4 * N], pkj[ks * 4 * LK], a1[q]);
                    a2[q] = mfma_f64(pki[ks * 4 * LK], pgj[ks * 4 * LG], a2[q]);
                }
            }
            wg_barrier();   // every operand read of G is done: a2 goes there for the transposed read
            double W[C::TPW5][4];
#pragma unroll
            for (int q = 0; q < C::TPW5; ++q) {
                const int tl = wave + 4 * q;
                const int tc = min(tl, C::NT5 - 1);
                const int it = tc / T_NP, jt = tc - it * T_NP;
                const int j = 16 * jt + c16;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 16 * it + g + 4 * v;
                    W[q][v] = 0.0;
                    if (tl < C::NT5) {
                        if (i < NP && j < N) sMt[i * LM + j] = a2[q][v];
                        if (i < N && j <= N) {
                            W[q][v] = (sQ[i * LQ + j] + a1[q][v]) + a2[q][v];
                            if (j < N) sQ[i * LQ + j] = W[q][v];
                        }
                    }
                }
            }
            wg_barrier();
            // ---- S6: V = W + a2^T ; V^T from the mirrored pair (same operands, same order) ; P <- (V + V^T)/2
#pragma unroll
            for (int q = 0; q < C::TPW5; ++q) {
                const int tl = wave + 4 * q;
                const int tc = min(tl, C::NT5 - 1);
                const int it = tc / T_NP, jt = tc - it * T_NP;
                const int j = 16 * jt + c16;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 16 * it + g + 4 * v;
                    if (tl < C::NT5 && i < N && j <= N) {
                        const double V = W[q][v] + sMt[j * LM + i];
                        double out = V;
                        if (j < N) {
                            const double Vt = sQ[j * LQ + i] + a2[q][v];
                            out = 0.5 * (V + Vt);
                        }
                        sP[i * LP + j] = out;
                    }
                }
            }
        }
        wg_barrier();
    }
    if (singular && sing && tid == 0) singular[b] = 1;
}

}  // namespace dpilqr
