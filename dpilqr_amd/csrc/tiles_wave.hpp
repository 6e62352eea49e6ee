// tiles_wave.hpp -- K1 specialised: the solve loop's tile producer for a batch of ONE model, compiled per
// (model, agents).  Dispatched (dpilqr_hip.hip) for linear models whose A, B, L_uu are already in place (DYN_ONLY) and
// up to 6 agents -- where it was measured faster than the generic producer; the !DYN_ONLY path is kept compilable.
//
// Same records, same values, same summation orders as k_make_tiles<NS,NC,true> (tiles.hpp: MultiDynamicalModel.
// linearize dynamics.py:173-186, GameCost.quadraticize cost.py:208-239, ProximityCost.quadraticize cost.py:135-171).
// The generic producer gives every (item, step) record its own wavefront, most of whose lanes idle through the pair
// derivatives and whose instruction stream is index arithmetic and per-entry parameter loads.  Here a wavefront owns a
// run of consecutive records of one item (the solve loop launches one group of RPG records per wavefront):
//   * Q + Q^T, Q_f + Q_f^T, R + R^T, x_f, n_dims are staged in LDS once per wavefront;
//   * records are processed in groups of RPG so that the fp64 sqrt / divisions of the pair derivatives of several
//     records run side by side (cfg2: 3 records x 10 pairs = 30 lanes);
//   * L_xx is written as whole rows in 16-byte pieces (its zeros included): full lines to HBM instead of the generic
//     kernel's scattered 8-byte entries; L_x, L_u and -- unless DYN_ONLY, see tiles.hpp -- the agents' A, B, L_uu blocks
//     follow;
//   * stores are issued behind the compiler's back (riccati_tiled.hpp) so the next group's loads are never made
//     to wait for them by count.
#pragma once
#include <hip/hip_runtime.h>

#include "riccati_tiled.hpp"
#include "tiles.hpp"

namespace dpilqr {

// Streaming ("nt") stores: the records are read next by a different kernel from every XCD, so there is nothing to
// gain from leaving them dirty in the writing XCD's L2.  Measured on the bench (tiles + the sweep that follows, ms per
// step): plain stores 0.347 + 0.860, write-through (sc0 sc1) 0.349 + 0.875, nt 0.392 + 0.819 -- the sweep gets its
// 10 % back; scattered 8-byte entries instead of whole L_xx rows are worse under every policy.
#define DPILQR_K1_STORE_MOD " nt"
__device__ __forceinline__ void k1_store_v2d(double* p, v2d v) {
    asm volatile("global_store_dwordx4 %0, %1, off" DPILQR_K1_STORE_MOD "\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void k1_store_f64(double* p, double v) {
    asm volatile("global_store_dwordx2 %0, %1, off" DPILQR_K1_STORE_MOD "\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}

template <int MODEL, int KA, bool DYN_ONLY>
struct TilesWaveCfg {
    static constexpr int NS = ModelDef<MODEL>::NS, NC = ModelDef<MODEL>::NC;
    static constexpr int n = KA * NS, m = KA * NC, NP = KA * (KA - 1) / 2, NP1 = NP > 0 ? NP : 1;
    static constexpr int widest = NP1 > KA ? NP1 : KA;
    // records per group = per wavefront: enough to fill the lanes of the pair derivatives for one or two agents; from
    // six pairs on, three records measured best (cfg2: K1 0.246 -> 0.230 ms per bench step against six, 0.241 with two,
    // 0.244 with four) -- more wavefronts in flight beat fuller lanes
    static constexpr int RPG_CAP = widest >= 6 ? 3 : 8;
    static constexpr int RPG = widest >= 64 ? 1 : (64 / widest > RPG_CAP ? RPG_CAP : 64 / widest);
    // per record: e = x - xf [n], u [m], pair gradients [NP1][3], pair Hessians [NP1][9], diagonal sums [KA][9], (A, B blocks)
    static constexpr int oE = 0, oU = oE + n, oG = oU + m, oH = oG + NP1 * 3, oD = oH + NP1 * 9, oA = oD + KA * 9;
    static constexpr int oB = oA + (DYN_ONLY ? 0 : KA * NS * NS);
    static constexpr int per_rec = (oB + (DYN_ONLY ? 0 : KA * NS * NC) + 1) & ~1;
    // per wavefront: QQ [KA][NS*NS], QQf likewise, RR [KA][NC*NC], xf [n]
    static constexpr int oQQ = RPG * per_rec, oQQf = oQQ + KA * NS * NS, oRR = oQQf + KA * NS * NS, oXf = oRR + KA * NC * NC;
    static constexpr int total = (oXf + n + 1) & ~1;
};

template <int MODEL, int KA, bool DYN_ONLY>
__global__ __launch_bounds__(64) void k_make_tiles_wave(dpilqr_batch_desc D, const double* __restrict__ X,
                                                         const double* __restrict__ U, double* __restrict__ tiles,
                                                         const int32_t* __restrict__ items,
                                                         const int32_t* __restrict__ n_items, int groups_per_wave,
                                                         int xx_rows) {
    using C = TilesWaveCfg<MODEL, KA, DYN_ONLY>;
    constexpr int NS = C::NS, NC = C::NC, n = C::n, m = C::m, NPAIRS = C::NP, NP1 = C::NP1, RPG = C::RPG;
    constexpr int PD = NS < 3 ? NS : 3;
    const int slot = blockIdx.y;
    if (n_items && slot >= *n_items) return;
    const int b = items ? items[slot] : slot;
    const int T = D.T, lane = threadIdx.x;
    const int n_groups = (T + 1 + RPG - 1) / RPG;
    const int g_first = blockIdx.x * groups_per_wave;
    if (g_first >= n_groups) return;
    const int g_last = min(g_first + groups_per_wave, n_groups);
    const ItemParams P = item_params(D, b);
    const TileLayout L(n, m);
    const double wr = D.w_ref, wp = D.w_prox;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sQQ = lds + C::oQQ;
    double* sQQf = lds + C::oQQf;
    double* sRR = lds + C::oRR;
    double* sXf = lds + C::oXf;

    // ---- once per wavefront: symmetrised weights, goals
    for (int e = lane; e < KA * NS * NS; e += 64) {
        const int a = e / (NS * NS), r = e - a * NS * NS, li = r / NS, lj = r - li * NS;
        sQQ[e] = P.Q[a * NS * NS + li * NS + lj] + P.Q[a * NS * NS + lj * NS + li];
        sQQf[e] = P.Qf[a * NS * NS + li * NS + lj] + P.Qf[a * NS * NS + lj * NS + li];
    }
    for (int e = lane; e < KA * NC * NC; e += 64) {
        const int a = e / (NC * NC), r = e - a * NC * NC, li = r / NC, lj = r - li * NC;
        sRR[e] = P.R[a * NC * NC + li * NC + lj] + P.R[a * NC * NC + lj * NC + li];
    }
    for (int e = lane; e < n; e += 64) sXf[e] = P.xf[e];
    const double radius = P.radius;
    DPILQR_LDS_FENCE();

    const double* Xb = X + (int64_t)b * (T + 1) * n;
    const double* Ub = U + (int64_t)b * T * m;
    double* recs = tiles + (int64_t)slot * (T + 1) * L.stride;   // records are indexed by list position

    for (int grp = g_first; grp < g_last; ++grp) {
        const int t0 = grp * RPG;
        const int n_rec = min(RPG, T + 1 - t0);
        // ---- phase A1: (record, agent): error vector, controls, (Jacobians)
        for (int idx = lane; idx < n_rec * KA; idx += 64) {
            const int r = idx / KA, a = idx - r * KA, t = t0 + r;
            const bool terminal = (t == T);
            double* sr = lds + r * C::per_rec;
            double x[NS], u[NC];
#pragma unroll
            for (int i = 0; i < NS; ++i) x[i] = Xb[(int64_t)t * n + a * NS + i];
#pragma unroll
            for (int i = 0; i < NC; ++i) u[i] = terminal ? 0.0 : Ub[(int64_t)t * m + a * NC + i];
#pragma unroll
            for (int i = 0; i < NS; ++i) sr[C::oE + a * NS + i] = x[i] - sXf[a * NS + i];
#pragma unroll
            for (int i = 0; i < NC; ++i) sr[C::oU + a * NC + i] = u[i];
            if constexpr (!DYN_ONLY) {
                if (!terminal) {
                    double A[NS * NS], Bm[NS * NC];
                    linearize<MODEL>(x, u, D.dt, A, Bm);
#pragma unroll
                    for (int i = 0; i < NS * NS; ++i) sr[C::oA + a * NS * NS + i] = A[i];
#pragma unroll
                    for (int i = 0; i < NS * NC; ++i) sr[C::oB + a * NS * NC + i] = Bm[i];
                }
            }
        }
        // ---- phase A2: (record, pair): proximity derivatives, pairs in itertools.combinations order
        for (int idx = lane; idx < n_rec * NPAIRS; idx += 64) {
            const int r = idx / NP1, p = idx - r * NP1, t = t0 + r;
            int i = 0, rem = p;
            while (rem >= KA - 1 - i) { rem -= KA - 1 - i; ++i; }
            const int j = i + 1 + rem;
            double* sr = lds + r * C::per_rec;
            const int nd = min(P.n_dims[i], P.n_dims[j]);   // cost.py:145
            double g[3], H[9];
            pair_quadraticize(Xb + (int64_t)t * n + i * NS, Xb + (int64_t)t * n + j * NS, nd, radius, g, H);
#pragma unroll
            for (int c = 0; c < 3; ++c) sr[C::oG + p * 3 + c] = g[c];
#pragma unroll
            for (int c = 0; c < 9; ++c) sr[C::oH + p * 9 + c] = H[c];
        }
        DPILQR_LDS_FENCE();
        // ---- phase A3: (record, agent, entry): the diagonal blocks' sums over the pairs containing the agent
        if (KA > 1) {
            for (int idx = lane; idx < n_rec * KA * 9; idx += 64) {
                const int r = idx / (KA * 9), q = idx - r * (KA * 9), a = q / 9, c = q - a * 9;
                double* sr = lds + r * C::per_rec;
                double acc = 0.0;
                for (int o = 0; o < KA; ++o) {
                    if (o == a) continue;
                    const int p = (o < a) ? pair_index(o, a, KA) : pair_index(a, o, KA);
                    acc += sr[C::oH + p * 9 + c];
                }
                sr[C::oD + q] = acc;
            }
            DPILQR_LDS_FENCE();
        }
        {   // ---- phase B1: L_xx = w_ref blockdiag(Q+Q^T) + w_prox sum_pairs(+-H), whole rows in 16-byte pieces
            constexpr int XE = (n % 2 == 0) ? 2 : 1;
            // only the first xx_rows rows of every agent's block row are written (all NS of them unless the caller
            // knows the others hold w_ref (Q + Q^T) only and has put them in place, see the dispatch)
            const int per = KA * xx_rows * (n / XE);
            for (int idx = lane; idx < n_rec * per; idx += 64) {
                const int r = idx / per, w = idx - r * per, t = t0 + r;
                const double* sr = lds + r * C::per_rec;
                const double* QQ = (t == T) ? sQQf : sQQ;
                const int rr = w / (n / XE), j0 = XE * (w - rr * (n / XE));
                const int ai = rr / xx_rows, li = rr - ai * xx_rows;
                const int i = ai * NS + li, e = i * n + j0;
                double v[XE];
#pragma unroll
                for (int c = 0; c < XE; ++c) {
                    const int j = j0 + c, aj = j / NS, lj = j - aj * NS;
                    double val = 0.0;
                    if (ai == aj) val = wr * QQ[ai * NS * NS + li * NS + lj];
                    if (KA > 1 && li < 3 && lj < 3) {
                        double acc = 0.0;
                        if (ai == aj) {
                            acc = sr[C::oD + ai * 9 + li * 3 + lj];
                        } else {
                            const int p = (ai < aj) ? pair_index(ai, aj, KA) : pair_index(aj, ai, KA);
                            acc += -sr[C::oH + p * 9 + li * 3 + lj];
                        }
                        val += wp * acc;
                    }
                    v[c] = val;
                }
                double* dst = recs + (int64_t)t * L.stride + L.oLxx + e;
                if constexpr (XE == 2) k1_store_v2d(dst, v2d{v[0], v[1]}); else k1_store_f64(dst, v[0]);
            }
        }
        // ---- phase B2: L_x = w_ref e^T (Q+Q^T) + w_prox sum_pairs(+-g) ; L_u = w_ref u^T (R+R^T)
        for (int idx = lane; idx < n_rec * n; idx += 64) {
            const int r = idx / n, j = idx - r * n, t = t0 + r;
            const double* sr = lds + r * C::per_rec;
            const double* QQ = (t == T) ? sQQf : sQQ;
            const int a = j / NS, lj = j - a * NS;
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) v += sr[C::oE + a * NS + i] * QQ[a * NS * NS + i * NS + lj];
            v = wr * v;
            if (KA > 1 && lj < 3) {
                double acc = 0.0;
                for (int o = 0; o < KA; ++o) {
                    if (o == a) continue;
                    if (o < a) acc += -sr[C::oG + pair_index(o, a, KA) * 3 + lj];
                    else       acc += sr[C::oG + pair_index(a, o, KA) * 3 + lj];
                }
                v += wp * acc;
            }
            k1_store_f64(recs + (int64_t)t * L.stride + L.oLx + j, v);
        }
        for (int idx = lane; idx < n_rec * m; idx += 64) {
            const int r = idx / m, j = idx - r * m, t = t0 + r;
            if (t == T) continue;
            const double* sr = lds + r * C::per_rec;
            const int a = j / NC, lj = j - a * NC;
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < NC; ++i) v += sr[C::oU + a * NC + i] * sRR[a * NC * NC + i * NC + lj];
            k1_store_f64(recs + (int64_t)t * L.stride + L.oLu + j, wr * v);
        }
        // ---- phase B3: what does not depend on (X, U) for a linear model with a shared R
        if constexpr (!DYN_ONLY) {
            for (int idx = lane; idx < n_rec * KA * NS * NS; idx += 64) {
                const int r = idx / (KA * NS * NS), e = idx - r * (KA * NS * NS), t = t0 + r;
                if (t == T) continue;
                const int a = e / (NS * NS), q = e - a * NS * NS, li = q / NS, lj = q - li * NS;
                k1_store_f64(recs + (int64_t)t * L.stride + L.oA + (a * NS + li) * L.ldAB + a * NS + lj,
                             lds[r * C::per_rec + C::oA + e]);
            }
            for (int idx = lane; idx < n_rec * KA * NS * NC; idx += 64) {
                const int r = idx / (KA * NS * NC), e = idx - r * (KA * NS * NC), t = t0 + r;
                if (t == T) continue;
                const int a = e / (NS * NC), q = e - a * NS * NC, li = q / NC, lj = q - li * NC;
                k1_store_f64(recs + (int64_t)t * L.stride + L.oB + (a * NS + li) * L.ldAB + a * NC + lj,
                             lds[r * C::per_rec + C::oB + e]);
            }
            for (int idx = lane; idx < n_rec * KA * NC * NC; idx += 64) {
                const int r = idx / (KA * NC * NC), e = idx - r * (KA * NC * NC), t = t0 + r;
                if (t == T) continue;
                const int a = e / (NC * NC), q = e - a * NC * NC, li = q / NC, lj = q - li * NC;
                k1_store_f64(recs + (int64_t)t * L.stride + L.oLuu + (a * NC + li) * L.ldUG + a * NC + lj, wr * sRR[e]);
            }
        }
        DPILQR_LDS_FENCE();
    }
}

}  // namespace dpilqr
