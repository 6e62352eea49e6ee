// forward_team.hpp -- K3 for launches of at most one item per SIMD (a job's draining tail, a single small batch): the line search
// of k_linesearch_wave (forward_wave.hpp) with a TEAM of two wavefronts per item.
//
// There the line search is a 50-step dependent chain of one wavefront per item, and what a step costs is its instruction count:
// x[t] -> dx -> du = K dx + alpha d -> u[t] -> RK4 -> x[t+1].  The costs of the step -- reference cost, pair costs, their sums in
// the reference's order: a third of the step's instructions -- feed only J, not the chain.  So the second wavefront takes them:
//   wavefront 0 (the rollout)   stages K[t], d[t], dx, x'[t] in LDS, forms u'[t], hands (x'[t], u'[t]) over in one of two LDS
//                               buffers, integrates, stores the candidates' trajectories;
//   wavefront 1 (the costs)     one step behind: reference cost and pair costs of (x'[t], u'[t]) per (candidate, agent) lane, the
//                               stage sums by the candidates' first lanes, J accumulated in time order; the terminal cost.
// One s_barrier per step (the buffer of step t is written before barrier t and read after it; it is written again in step t + 2,
// after barrier t + 1, which the cost wavefront reaches only when it has finished step t).  The same expressions on the same
// values in the same order: costs, decisions, trajectories bit-identical to k_linesearch_wave
// (tests/test_gpu_parity.py::test_line_search_team_equals_the_one_wavefront_line_search).  For the sizes with one wavefront per
// item (k n_alpha <= 64 lanes: up to six agents); instantiated for the four-state models, QuadcopterDynamics6D, DoubleIntDynamics6D and CarDynamics3D.
#pragma once
#include <hip/hip_runtime.h>

#include "forward_wave.hpp"

namespace dpilqr {

template <int MODEL, int KA>
struct TeamFwdLds {
    using W = WaveFwdLds<MODEL, KA>;
    static constexpr int NS = W::NS, NC = W::NC, n = W::n, m = W::m, NP1 = W::NP1, LDG = W::LDG, NG = DPILQR_N_ALPHA;
    static constexpr int LDU = (m % 4 == 0) ? m + 2 : m;           // a candidate's row of u' (see WaveFwdLds::LDG)
    static constexpr int oK = 0;                                   // K[t]  m x n
    static constexpr int od = oK + m * n;                          // d[t]  m
    static constexpr int odx = (od + m + 1) & ~1;                  // dx    [g][LDG]
    static constexpr int oxs = odx + NG * LDG;                     // x'    [2][g][LDG]
    static constexpr int ous = oxs + 2 * NG * LDG;                 // u'    [2][g][LDU]
    static constexpr int ocr = (ous + 2 * NG * LDU + 1) & ~1;      // ref cost  [g][KA]
    static constexpr int ocp = ocr + NG * KA;                      // pair cost [g][NP1]
    static constexpr int oJ = ocp + NG * NP1;                      // J [g]
    static constexpr int octl = oJ + NG;
    static constexpr int total = (octl + 2 + 1) & ~1;
    static constexpr bool supported = W::NW == 1;      // (the per-agent constants live in the cost wavefront's registers: it has few others)
};

__device__ __forceinline__ void team_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void team_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODEL, int KA>
__global__ __launch_bounds__(128, 2) void k_linesearch_team(
    dpilqr_batch_desc D, double* X, double* U, const double* __restrict__ K, const double* __restrict__ d,
    const double* __restrict__ alphas, double* Xc, double* Uc, SolveState S, const int32_t* __restrict__ items,
    const int32_t* __restrict__ n_items) {
    using W = TeamFwdLds<MODEL, KA>;
    static_assert(W::supported, "team line search: one wavefront per item");
    constexpr int NS = W::NS, NC = W::NC, n = W::n, m = W::m, mn = m * n, NPAIRS = KA * (KA - 1) / 2, NP1 = W::NP1;
    constexpr int NG = DPILQR_N_ALPHA, LDG = W::LDG, LDU = W::LDU;
    constexpr int PPL = (NPAIRS + KA - 1) / KA;
    constexpr int KV = (mn / 2 + 63) / 64;

    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0: the rollout, 1: the costs
    const int tid = (int)(threadIdx.x & 63);
    const int slot = blockIdx.x;
    if (slot >= *n_items) return;            // (both wavefronts of the item: no barrier is left waiting)
    const int b = items[slot];
    const int T = D.T;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* Xb = X + (int64_t)b * (T + 1) * n;
    double* Ub = U + (int64_t)b * T * m;
    if (S.singular && S.singular[b]) {  // np.linalg.solve would have raised LinAlgError
        if (threadIdx.x == 0) retire_without_gains(S, b);
        return;
    }
    const bool active = tid < KA * NG;
    const int g = active ? tid / KA : 0, a = active ? tid - (tid / KA) * KA : 0;
    double* sxs0 = lds + W::oxs + g * LDG;           // + parity * NG * LDG
    double* sus0 = lds + W::ous + g * LDU;           // + parity * NG * LDU

    if (role == 0) {
        // ================= the rollout =================
        const int64_t gslot = S.gains_by_item ? b : slot;
        const double* Kb = K + gslot * T * mn;
        const double* db = d + gslot * T * m;
        const double alpha = alphas[g];
        double* Xw = Xc + ((int64_t)slot * NG + g) * (int64_t)(T + 1) * n + a * NS;
        double* Uw = Uc + ((int64_t)slot * NG + g) * (int64_t)T * m + a * NC;
        double* sK = lds + W::oK;
        double* sd = lds + W::od;
        double* sdx = lds + W::odx + g * LDG;
        v2d stK[KV];
        double std_ = 0.0, u[NC], xold[NS], x[NS];
        auto fetch = [&](int t) {       // K[t] last: see k_linesearch_wave
#pragma unroll
            for (int i = 0; i < NS; ++i) xold[i] = Xb[(int64_t)t * n + a * NS + i];
#pragma unroll
            for (int i = 0; i < NC; ++i) u[i] = Ub[(int64_t)t * m + a * NC + i];
            std_ = db[(int64_t)t * m + min(tid, m - 1)];
            const double* Kt = Kb + (int64_t)t * mn;
#pragma unroll
            for (int q = 0; q < KV; ++q) {
                const int e = min(tid + 64 * q, mn / 2 - 1);
                stK[q] = *reinterpret_cast<const v2d*>(Kt + 2 * e);
            }
        };
        auto store_vec = [&](double* p, const double* v, int len) {
            if ((len & 1) == 0) {
#pragma unroll
                for (int i = 0; i < len; i += 2) store_v2d_nt(p + i, v2d{v[i], v[i + 1]});
            } else {
#pragma unroll
                for (int i = 0; i < len; ++i) store_f64_nt(p + i, v[i]);
            }
        };
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = Xb[a * NS + i];
        fetch(0);
        double ut[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) ut[i] = 0.0;
        for (int t = 0; t < T; ++t) {
            double* sxs = sxs0 + (t & 1) * NG * LDG;
            double* sus = sus0 + (t & 1) * NG * LDU;
#pragma unroll
            for (int q = 0; q < KV; ++q) {
                const int e = min(tid + 64 * q, mn / 2 - 1);
                *reinterpret_cast<v2d*>(sK + 2 * e) = stK[q];
            }
            if (tid < m) sd[tid] = std_;
            if (active) {
                store_vec(Xw + (int64_t)t * n, x, NS);
                if (t > 0) store_vec(Uw + (int64_t)(t - 1) * m, ut, NC);
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) ut[i] = u[i];
            if (active) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    sdx[a * NS + i] = x[i] - xold[i];   // dx = X'[t] - X[t]
                    sxs[a * NS + i] = x[i];
                }
            }
            DPILQR_LDS_FENCE();
            if (t + 1 < T) fetch(t + 1);
            DPILQR_LDS_FENCE();
            {   // du = K[t] dx + alpha d[t] (control.py:106): this agent's NC rows, j ascending
                double sum[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) sum[c] = 0.0;
                const double* rows = sK + a * (NC * n);
                if (n % 2 == 0) {
#pragma unroll 5
                    for (int j = 0; j < n; j += 2) {
                        const v2d dx2 = *reinterpret_cast<const v2d*>(lds + W::odx + g * LDG + j);
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const v2d kr = *reinterpret_cast<const v2d*>(rows + c * n + j);
                            sum[c] += kr.x * dx2.x;
                            sum[c] += kr.y * dx2.y;
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < n; ++j) {
                        const double dxj = lds[W::odx + g * LDG + j];
#pragma unroll
                        for (int c = 0; c < NC; ++c) sum[c] += rows[c * n + j] * dxj;
                    }
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const double du = sum[c] + alpha * sd[a * NC + c];
                    ut[c] = ut[c] + du;
                }
            }
            if (active) {
#pragma unroll
                for (int c = 0; c < NC; ++c) sus[a * NC + c] = ut[c];
            }
            double xn[NS];
            integrate<MODEL>(x, ut, D.dt, xn);
#pragma unroll
            for (int i = 0; i < NS; ++i) x[i] = xn[i];
            team_barrier_lds();     // (x'[t], u'[t]) are the cost wavefront's; it has finished step t - 1
        }
        if (active) {
            store_vec(Xw + (int64_t)T * n, x, NS);
            if (T > 0) store_vec(Uw + (int64_t)(T - 1) * m, ut, NC);
#pragma unroll
            for (int i = 0; i < NS; ++i) sxs0[(T & 1) * NG * LDG + a * NS + i] = x[i];
        }
        team_barrier();             // x'[T] handed over; the candidates' trajectory stores have landed
        team_barrier_lds();         // J is in place
        int* ctl = reinterpret_cast<int*>(lds + W::octl);
        if (tid == 0) ctl[0] = linesearch_decide(S, b, NG, lds + W::oJ);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        team_barrier_lds();
    } else {
        // ================= the costs =================
        const ItemParams P = item_params(D, b);
        double xf[NS], Q[NS * NS], R[NC * NC];
#pragma unroll
        for (int i = 0; i < NS; ++i) xf[i] = P.xf[a * NS + i];
#pragma unroll
        for (int i = 0; i < NS * NS; ++i) Q[i] = P.Q[a * NS * NS + i];
#pragma unroll
        for (int i = 0; i < NC * NC; ++i) R[i] = P.R[a * NC * NC + i];
        const double radius = P.radius;
        bool homog = true;
#pragma unroll
        for (int i = 1; i < KA; ++i) homog = homog && (P.n_dims[i] == P.n_dims[0]);
        int pp[PPL > 0 ? PPL : 1];   // this lane's pairs, p = a + q * KA: i | j << 8 | n_dims << 16
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int p = min(a + q * KA, NP1 - 1);
            int ii = 0, rem = p;
            while (rem >= KA - 1 - ii) { rem -= KA - 1 - ii; ++ii; }
            const int jj = ii + 1 + rem;
            pp[q] = ii | (jj << 8) | ((homog ? 2 : min(P.n_dims[ii], P.n_dims[jj])) << 16);
        }
        double* scr = lds + W::ocr + g * KA;
        double* scp = lds + W::ocp + g * NP1;
        double J = 0.0;
        auto stage = [&](int t, bool terminal, const double* Mq) {   // the cost of (x'[t], u'[t]); summed in the reference's order
            const double* sxs = sxs0 + (t & 1) * NG * LDG;
            const double* sus = sus0 + (t & 1) * NG * LDU;
            double x[NS], uu[NC];
#pragma unroll
            for (int i = 0; i < NS; ++i) x[i] = sxs[a * NS + i];
#pragma unroll
            for (int c = 0; c < NC; ++c) uu[c] = terminal ? 0.0 : sus[a * NC + c];
            const double cr = ref_cost<NS, NC>(x, uu, xf, Mq, R, terminal);
            double cp[PPL > 0 ? PPL : 1];
#pragma unroll
            for (int q = 0; q < PPL; ++q) cp[q] = pair_cost_nd<NS>(sxs + (pp[q] & 255) * NS, sxs + ((pp[q] >> 8) & 255) * NS, pp[q] >> 16, radius);
            if (active) {
                scr[a] = cr;
#pragma unroll
                for (int q = 0; q < PPL; ++q)
                    if (a + q * KA < NPAIRS) scp[a + q * KA] = cp[q];
            }
            DPILQR_LDS_FENCE();
            if (a == 0) {
                double prox = 0.0, ref = 0.0;
#pragma unroll
                for (int p = 0; p < NPAIRS; ++p) prox += scp[p];
#pragma unroll
                for (int i = 0; i < KA; ++i) ref += scr[i];
                J += D.w_prox * prox + D.w_ref * ref;
            }
            DPILQR_LDS_FENCE();
        };
        for (int t = 0; t < T; ++t) {
            team_barrier_lds();     // step t's (x', u') are in buffer t & 1
            stage(t, false, Q);
        }
        team_barrier();             // x'[T]
        {
            double Qf[NS * NS];
#pragma unroll
            for (int i = 0; i < NS * NS; ++i) Qf[i] = P.Qf[a * NS * NS + i];
            stage(T, true, Qf);     // cost(X[T], 0, terminal=True) (control.py:112)
        }
        if (active && a == 0) lds[W::oJ + g] = J;
        team_barrier_lds();         // J is in place
        team_barrier_lds();         // the decision
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const int acc = reinterpret_cast<const int*>(lds + W::octl)[0];
    if (acc < 0) return;
    // accepted: X, U <- the accepted candidate's trajectory (a coalesced copy out of the scratch, both wavefronts)
    const double* Xa = Xc + ((int64_t)slot * NG + acc) * (int64_t)(T + 1) * n;
    const double* Ua = Uc + ((int64_t)slot * NG + acc) * (int64_t)T * m;
    for (int e = (int)threadIdx.x; e < (T + 1) * n; e += 128) Xb[e] = Xa[e];
    for (int e = (int)threadIdx.x; e < T * m; e += 128) Ub[e] = Ua[e];
}

}  // namespace dpilqr
