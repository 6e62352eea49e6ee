// forward_wave.hpp -- K3 specialised: the line-searched forward pass of one solver iteration with ONE wavefront
// per sub-problem, agents and model known at compile time.
//
// Reference: ilqrSolver._forward_pass (control.py:95-114) for the ten alphas of control.py:162 and the accept /
// regularisation logic of the iteration loop (control.py:179-211) -- the same arithmetic, in the same order, as
// the generic k_forward (forward.hpp), which remains the path for heterogeneous models, large k and the API's
// stand-alone passes.  This kernel exists because the forward pass is a 50-step dependent chain per candidate:
// what it costs is the latency of one step, and the generic kernel spends most of a step waiting on memory
// (per-step parameter loads, a prefetch that is drained as soon as it is issued).  Here
//   * everything that does not change along the horizon (x_f, Q, R, radius, n_dims) is read once;
//   * K[t], d[t], X[t], U[t] of step t+1 are in flight while step t computes, and the candidate trajectories are
//     written with stores the compiler's wait-count bookkeeping does not see, so the loads are waited on by
//     count, never drained;
//   * the pair costs are dealt round-robin to the agents' lanes instead of triangularly;
//   * a single wavefront executes its LDS operations in order: no barriers, no double buffering.
// Lane (g, a) = (candidate g, agent a), tid = g * KA + a; KA * n_alpha <= 64.
#pragma once
#include <hip/hip_runtime.h>

#include "forward.hpp"
#include "riccati_tiled.hpp"   // v2d, store_v2d_nt

namespace dpilqr {

template <int KA>
struct PairTable {   // itertools.combinations(range(KA), 2) order
    int i[KA * (KA - 1) / 2 + 1], j[KA * (KA - 1) / 2 + 1];
    constexpr PairTable() : i{}, j{} {
        int p = 0;
        for (int a = 0; a < KA; ++a)
            for (int b = a + 1; b < KA; ++b) { i[p] = a; j[p] = b; ++p; }
    }
};

// The accept / regularisation decision of one iteration for item b given the candidates' costs in sJ[0..ngrp)
// (control.py:179-211, _decrease_regularization :232-237).  Called by ONE thread.  Returns the accepted
// candidate or -1.
__device__ inline int linesearch_decide(const SolveState& S, int b, int ngrp, const double* sJ) {
    const int iter = S.n_bwd[b];   // this item's own iteration index (items join the batch at different times)
    const double J_star = S.J_star[b];
    int acc = -1;
    for (int i = 0; i < ngrp; ++i)
        if (sJ[i] < J_star) { acc = i; break; }  // strict <, NaN rejects (control.py:183)
    const int n_eval = (acc >= 0) ? acc + 1 : ngrp;
    const double J_last = sJ[n_eval - 1];         // last EVALUATED cost (quirk Q2)
    const double mu_before = S.mu[b];
    int status = DPILQR_STATUS_ACTIVE;
    double J_new = J_star;
    if (acc >= 0) {
        const bool conv = fabs((J_star - J_last) / J_star) < S.tol;  // control.py:184
        J_new = J_last;
        double delta = fmin(1.0, S.delta[b]) / 2.0;
        double mu = mu_before * delta;
        if (mu <= 1e-6) mu = 0.0;
        S.delta[b] = delta; S.mu[b] = mu; S.J_star[b] = J_new;
        if (conv) status = DPILQR_STATUS_CONVERGED;
        else if (solve_time_is_up(S, b)) status = DPILQR_STATUS_KILLED;   // control.py:213-218, checked before the loop bound
        else if (iter + 1 >= S.n_lqr_iter) status = DPILQR_STATUS_MAX_ITER;
    } else {
        status = DPILQR_STATUS_LINESEARCH_FAILED;  // control.py:195-198
    }
    S.J_last[b] = J_last;
    S.n_fwd[b] += n_eval;
    S.n_bwd[b] = iter + 1;
    S.status[b] = status;
    if (S.trace) {
        double* tr = S.trace + ((int64_t)b * S.n_lqr_iter + iter) * 5;
        tr[0] = mu_before; tr[1] = (double)acc; tr[2] = J_last; tr[3] = J_new; tr[4] = (double)n_eval;
    }
    if (status == DPILQR_STATUS_ACTIVE && S.next_items) {
        const int pos = atomicAdd(S.next_count, 1);
        S.next_items[pos] = b;
    }
    return acc;
}

// pair_cost (cost.hpp) with the coordinate loop resolved: nd is 2 or 3 (ProximityCost.n_dims, cost.py:111)
template <int NS>
__device__ __forceinline__ double pair_cost_nd(const double* a, const double* b, int nd, double radius) {
    const double dx = a[0] - b[0], dy = a[1] - b[1];
    double s = 0.0;
    s += dx * dx;
    s += dy * dy;
    if (NS >= 3) {
        const double dz = a[2] - b[2];
        const double s3 = s + dz * dz;
        s = (nd >= 3) ? s3 : s;
    }
    if (nd < 2) s = dx * dx;   // never the case for the reference's models; kept for exactness with the loop
    if (s > radius * radius * (1.0 + 1e-12)) return 0.0;
    const double m = fmin(0.0, sqrt(s) - radius);
    return m * m;
}

template <int MODEL, int KA>
struct WaveFwdLds {
    static constexpr int NS = ModelDef<MODEL>::NS, NC = ModelDef<MODEL>::NC;
    static constexpr int n = KA * NS, m = KA * NC, NP = KA * (KA - 1) / 2, NP1 = NP > 0 ? NP : 1;
    // a candidate's row of dx / x': lanes of different candidates read the same column at once, so the row stride (in
    // 4-byte banks, mod 64) must not repeat among the 10 candidates: 20 doubles = 40 banks collides for candidates 0/8
    // and 1/9, 22 doubles does not -- any n that is a multiple of 4 collides.  Line search 0.361 -> 0.345 ms per bench
    // step.  (The same padding of K[t]'s rows, read by lanes of different agents, measured slower: 0.398 ms.)
    static constexpr int LDG = (n % 4 == 0) ? n + 2 : n;
    static constexpr int NW = (KA * DPILQR_N_ALPHA + 63) / 64;     // wavefronts per sub-problem
    // K[t]: the lanes of different AGENTS read their NC rows at the same column at once (16-byte reads), i.e. addresses
    // NC n doubles apart: at 15 unicycles that is 240 dwords = 48 mod 64 banks, so agents 0, 4, 8, 12 collide -- four-way, 3.9
    // conflict cycles per LDS instruction over the whole kernel (profiles/r03_wg_counters.csv).  A gap of KGAP doubles after
    // every agent's block of rows makes the agents' distance 4 x (an odd number) banks: sixteen agents, sixteen different
    // four-bank groups.  (Per-ROW padding measured slower at cfg2 in round 1; the per-agent gap keeps the rows of one agent
    // contiguous and costs the staging one precomputed offset per copied pair.)  Clusters of 7+ agents (two or three
    // wavefronts per item); DPILQR_LS_KGAP_ALL: A/B builds with the gap for every size.
#ifndef DPILQR_LS_KGAP_ALL
#define DPILQR_LS_KGAP_ALL 0
#endif
#ifndef DPILQR_LS_NO_KGAP
#define DPILQR_LS_NO_KGAP 0
#endif
    static constexpr int kgap_for(int blk) { int g = 0; while ((((blk + g) / 2) & 1) == 0) g += 2; return g; }
    static constexpr int KBLK = NC * n;                            // doubles of one agent's rows of K[t]
    static constexpr int KGAP = (!DPILQR_LS_NO_KGAP && (NW > 1 || DPILQR_LS_KGAP_ALL) && KBLK % 2 == 0) ? kgap_for(KBLK) : 0;
    static constexpr int oK = 0;                                   // K[t]  m x n (+ the gaps)
    static constexpr int od = oK + m * n + KA * KGAP;              // d[t]  m
    static constexpr int odx = (od + m + 1) & ~1;                  // dx    [g][n]
    static constexpr int oxs = odx + DPILQR_N_ALPHA * LDG;         // x'    [g][LDG]
    static constexpr int ocr = oxs + DPILQR_N_ALPHA * LDG;         // ref cost  [parity][g][KA]
    static constexpr int ocp = ocr + 2 * DPILQR_N_ALPHA * KA;      // pair cost [parity][g][NP1]
    static constexpr int oJ = ocp + 2 * DPILQR_N_ALPHA * NP1;      // J [g]
    static constexpr int IPB = NW == 1 ? 4 : 1;                    // sub-problems per workgroup
    static constexpr int octl = oJ + DPILQR_N_ALPHA;
    // items of more than one wavefront (7+ agents) and six- / twelve-state agents keep the per-agent constants Q, R, x_f in
    // LDS: in registers (48 for a four-state agent, 104 for a six-state one, next to the pair tables) those kernels spill
    // up to 550 bytes -- Quadcopter12D: 2 KB -- per lane into the horizon loop.  (One to six four-state agents: registers are
    // the faster place, measured.)
#ifndef DPILQR_LS_CL_ALL   // A/B builds: the per-agent constants in LDS for every size
#define DPILQR_LS_CL_ALL 0
#endif
    static constexpr bool CONST_LDS = NW > 1 || NS >= 6 || DPILQR_LS_CL_ALL;
    // Wavefronts per SIMD the register allocation aims at.  Two; one for nine and ten six-state agents, where two meant 32 / 63
    // spilled registers and 24 scratch loads per step INSIDE the horizon loop (ten quadcopters: 29.8 -> 21.0 ms of line search in a
    // 2048-item solve, nine: 18.8 -> 17.2; eight: equal, seven and the unicycle clusters of 12 .. 15: slower with one).  Three
    // (168 registers, 8 spilled) makes cfg2's five-agent kernel 20 % slower.  DPILQR_LS_OCC: A/B builds.
#ifndef DPILQR_LS_OCC
#define DPILQR_LS_OCC 2
#endif
    // ... and one for twelve-state agents (Quadcopter12D's RK4 stages hold 7 x 12 doubles next to the sincos expansions: 64 .. 102
    // scratch loads per step at two; one to three agents 11 .. 18 % faster with one, five equal; scripts/bench_q12.py)
    static constexpr int OCC = ((NW > 1 && NS >= 6 && KA >= 9) || NS >= 12) ? 1 : DPILQR_LS_OCC;
    // Steps of K[t], d[t], X[t], U[t] in flight ahead of the one being computed.  A step is a dependent chain of about the length of
    // one trip to HBM: with one step ahead the wavefront still waits for its data at the top of every step.  Two stages where the
    // registers allow it (four-state agents, one wavefront per item: 22 more registers of 50 spare).  DPILQR_LS_PF: A/B builds.
#ifndef DPILQR_LS_PF
#define DPILQR_LS_PF 2
#endif
    static constexpr int PF = (NW == 1 && NS == 4 && !CONST_LDS) ? DPILQR_LS_PF : 1;
    static_assert(PF == 1 || PF == 2, "the horizon loop is written out for one or two register stages (step t, step t + 1)");
    static constexpr int oQ = (octl + 2 + 1) & ~1;                 // Q [agent][NS*NS]
    static constexpr int oR = oQ + (CONST_LDS ? KA * NS * NS : 0); // R [agent][NC*NC]
    static constexpr int oXf = oR + (CONST_LDS ? KA * NC * NC : 0);
    static constexpr int total = (oXf + (CONST_LDS ? n : 0) + 1) & ~1;
};

// NW = 1 (k * 10 <= 64 lanes): four sub-problems per workgroup, one wavefront each, no barriers.  NW = 2, 3 (7..15
// agents): one sub-problem per workgroup; its wavefronts meet at two s_barriers per step (staged data written /
// everybody done reading it) and the per-step cost slots alternate by parity, because a candidate's lanes can
// straddle two wavefronts.
template <int NW>
__device__ __forceinline__ void wave_sync() {
    if constexpr (NW == 1) asm volatile("" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int MODEL, int KA>
__global__ __launch_bounds__((64 * WaveFwdLds<MODEL, KA>::NW * WaveFwdLds<MODEL, KA>::IPB), (WaveFwdLds<MODEL, KA>::OCC)) void k_linesearch_wave(
    dpilqr_batch_desc D, double* X, double* U, const double* __restrict__ K, const double* __restrict__ d,
    const double* __restrict__ alphas, double* Xc, double* Uc, SolveState S, const int32_t* __restrict__ items,
    const int32_t* __restrict__ n_items) {
    using W = WaveFwdLds<MODEL, KA>;
    constexpr int NS = W::NS, NC = W::NC, n = W::n, m = W::m, mn = m * n, NPAIRS = W::NP, NP1 = W::NP1;
    constexpr int NG = DPILQR_N_ALPHA;
    constexpr int NW = W::NW, NTH = 64 * NW;
    constexpr int PPL = (NPAIRS + KA - 1) / KA;          // pair costs per agent lane (round-robin deal)
    constexpr int KV = (mn / 2 + NTH - 1) / NTH;         // v2d of K[t] a lane stages (mn is even: NC*NS*KA*KA)

    const int sub = NW == 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int tid = NW == 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const int slot = blockIdx.x * W::IPB + sub;
    if (slot >= *n_items) return;
    const int b = items[slot];
    const int T = D.T;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    double* lds = lds_all + sub * W::total;
    double* Xb = X + (int64_t)b * (T + 1) * n;
    double* Ub = U + (int64_t)b * T * m;
    if (S.singular && S.singular[b]) {  // np.linalg.solve would have raised LinAlgError
        if (tid == 0) retire_without_gains(S, b);
        return;
    }
    const int64_t gslot = S.gains_by_item ? b : slot;
    const double* Kb = K + gslot * T * mn;
    const double* db = d + gslot * T * m;
    const bool active = tid < KA * NG;
    const int g = active ? tid / KA : 0, a = active ? tid - (tid / KA) * KA : 0;
    const double alpha = alphas[g];
    // this lane's candidate rows: a base that is the same in every lane (scalar registers) + a 32-bit lane offset -- as two 64-bit
    // per-lane pointers they were four of the registers the 14- and 15-agent kernels did not have
    double* const Xw0 = Xc + (int64_t)slot * NG * (int64_t)(T + 1) * n;
    double* const Uw0 = Uc + (int64_t)slot * NG * (int64_t)T * m;
    int xw_off = g * (T + 1) * n + a * NS;
    asm volatile("" : "+v"(xw_off));      // (a 32-bit value, not the 64-bit product it comes from)
    // (U's offset is formed where it is used, from g and a -- two multiply-adds per step instead of a register held, or spilled, for it)
    auto uw_off_at = [&](int t_) {
        int g_ = g;
        asm volatile("" : "+v"(g_));
        return (unsigned)(g_ * (T * m) + a * NC + t_ * m);
    };

    // ---- per-item constants, read once
    const ItemParams P = item_params(D, b);
    constexpr bool CL = W::CONST_LDS;
    double xf_r[CL ? 1 : NS], Q_r[CL ? 1 : NS * NS], R_r[CL ? 1 : NC * NC];
    const double* xf = xf_r; const double* Q = Q_r; const double* R = R_r;
    if constexpr (CL) {
        for (int e = tid; e < KA * NS * NS; e += NTH) lds[W::oQ + e] = P.Q[e];
        for (int e = tid; e < KA * NC * NC; e += NTH) lds[W::oR + e] = P.R[e];
        for (int e = tid; e < n; e += NTH) lds[W::oXf + e] = P.xf[e];
        xf = lds + W::oXf + a * NS; Q = lds + W::oQ + a * NS * NS; R = lds + W::oR + a * NC * NC;
    } else {
#pragma unroll
        for (int i = 0; i < NS; ++i) xf_r[i] = P.xf[a * NS + i];
#pragma unroll
        for (int i = 0; i < NS * NS; ++i) Q_r[i] = P.Q[a * NS * NS + i];
#pragma unroll
        for (int i = 0; i < NC * NC; ++i) R_r[i] = P.R[a * NC * NC + i];
    }
    const double radius = P.radius;
    bool homog = true;
#pragma unroll
    for (int i = 1; i < KA; ++i) homog = homog && (P.n_dims[i] == P.n_dims[0]);
    int pp[PPL > 0 ? PPL : 1];   // this lane's pairs, p = a + q * KA: i | j << 8 | n_dims << 16
    auto make_pairs = [&](int a_) {
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int p = min(a_ + q * KA, NP1 - 1);
            int ii = 0, rem = p;                              // p-th pair of itertools.combinations(range(KA), 2)
            while (rem >= KA - 1 - ii) { rem -= KA - 1 - ii; ++ii; }
            const int jj = ii + 1 + rem;
            pp[q] = ii | (jj << 8) | ((homog ? 2 : min(P.n_dims[ii], P.n_dims[jj])) << 16);
        }
    };
    make_pairs(a);

    double* sK = lds + W::oK;
    double* sd = lds + W::od;
    double* sdx = lds + W::odx + g * W::LDG;
    double* sxs = lds + W::oxs + g * W::LDG;
    double* scr0 = lds + W::ocr + g * KA;            // + parity * NG * KA
    double* scp0 = lds + W::ocp + g * NP1;           // + parity * NG * NP1

    // ---- step data: registers <- HBM, PF steps ahead (one register stage per step in flight)
    struct Stage { v2d stK[KV]; double std_, u[NC], xold[NS]; };
    constexpr int PF = W::PF;
    Stage stg[PF];
    double x[NS];
    // K[t] LAST: loads return in order and K[t] is what a step consumes first (the staging at its top), so the one wait of a step
    // stands there, where everything outstanding is at least a step old.  With X[t] last the wait for it stood behind the
    // candidate stores of the step and drained them too -- a store's round trip in every step of the chain.
    // (addresses: a base that is the same in every lane + an unsigned 32-bit element offset formed per step -- as per-lane 64-bit
    // pointers hoisted out of the horizon loop the four of them were eight registers, spilled and reloaded in every step by the
    // six- and eight-quadcopter kernels)
    auto fetch = [&](Stage& sg, int t) {
        const unsigned ko = (unsigned)(t * mn), xo = (unsigned)(t * n + a * NS), uo = (unsigned)(t * m + a * NC);
#ifdef DPILQR_LS_K_FIRST   // A/B builds: the order of rounds 1-3
#pragma unroll
        for (int q = 0; q < KV; ++q) {
            const int e = min(tid + NTH * q, mn / 2 - 1);
            sg.stK[q] = *reinterpret_cast<const v2d*>(Kb + (ko + 2u * (unsigned)e));
        }
#endif
#pragma unroll
        for (int i = 0; i < NS; ++i) sg.xold[i] = Xb[xo + (unsigned)i];
#pragma unroll
        for (int i = 0; i < NC; ++i) sg.u[i] = Ub[uo + (unsigned)i];
        sg.std_ = db[(unsigned)(t * m + min(tid, m - 1))];
#ifndef DPILQR_LS_K_FIRST
#pragma unroll
        for (int q = 0; q < KV; ++q) {
            const int e = min(tid + NTH * q, mn / 2 - 1);
            sg.stK[q] = *reinterpret_cast<const v2d*>(Kb + (ko + 2u * (unsigned)e));
        }
#endif
    };
    auto store_vec = [&](double* p, const double* v, int len) {   // len doubles, 16-byte aligned when len is even
        if ((len & 1) == 0) {
#pragma unroll
            for (int i = 0; i < len; i += 2) store_v2d_nt(p + i, v2d{v[i], v[i + 1]});
        } else {
#pragma unroll
            for (int i = 0; i < len; ++i) store_f64_nt(p + i, v[i]);
        }
    };
    auto stage_cost = [&](double& J, int par) {   // summed in the reference's order: pairs (combinations order), agents, time
        const double* scp = scp0 + par * NG * NP1;
        const double* scr = scr0 + par * NG * KA;
        double prox = 0.0, ref = 0.0;
        if constexpr (NPAIRS > 32) {
            // (in chunks: fully unrolled, the 105 pair costs of a fifteen-agent cluster were all loaded before the first add -- 210
            // registers for a moment, the source of the 27..29 spilled registers of the 14- and 15-agent kernels; the adds keep
            // their order)
            constexpr int CH = 8;
#pragma unroll 1
            for (int p0 = 0; p0 < NPAIRS; p0 += CH) {
                double v[CH];
#pragma unroll
                for (int q = 0; q < CH; ++q) v[q] = scp[min(p0 + q, NP1 - 1)];
#pragma unroll
                for (int q = 0; q < CH; ++q) prox = (p0 + q < NPAIRS) ? prox + v[q] : prox;
            }
        } else {
#pragma unroll
            for (int p = 0; p < NPAIRS; ++p) prox += scp[p];
        }
#pragma unroll
        for (int i = 0; i < KA; ++i) ref += scr[i];
        J += D.w_prox * prox + D.w_ref * ref;
    };
    auto post_costs = [&](double cr, const double* cp, int par) {
        if (active) {
            scr0[par * NG * KA + a] = cr;
#pragma unroll
            for (int q = 0; q < PPL; ++q)
                if (a + q * KA < NPAIRS) scp0[par * NG * NP1 + a + q * KA] = cp[q];
        }
    };

#pragma unroll
    for (int i = 0; i < NS; ++i) x[i] = Xb[a * NS + i];
#pragma unroll
    for (int s = 0; s < PF; ++s)
        if (s < T) fetch(stg[s], s);
    double J = 0.0;
    double ut[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) ut[i] = 0.0;

#ifdef DPILQR_PHASE_STAMPS
    unsigned long long w_wait = 0, w_t0 = __builtin_amdgcn_s_memtime();
#endif
    auto step = [&](Stage& sg, int t) {
        v2d (&stK)[KV] = sg.stK;
        double& std_ = sg.std_;
        double (&u)[NC] = sg.u;
        double (&xold)[NS] = sg.xold;
#ifdef DPILQR_PHASE_STAMPS
        {
            const unsigned long long a0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            w_wait += __builtin_amdgcn_s_memtime() - a0;
        }
#endif
        // K[t], d[t], dx, x' -> LDS
#pragma unroll
        for (int q = 0; q < KV; ++q) {
            // where this lane's pair goes: behind the gaps of the agents before it (WaveFwdLds::KGAP).  Formed here from a lane id
            // the optimiser cannot see through: hoisted out of the horizon loop these offsets were KV more registers in kernels
            // that already spill
            int tid_o = tid;
            if constexpr (W::KGAP > 0) asm volatile("" : "+v"(tid_o));
            const int e = min(tid_o + NTH * q, mn / 2 - 1);
            *reinterpret_cast<v2d*>(sK + 2 * e + (W::KGAP > 0 ? ((2 * e) / W::KBLK) * W::KGAP : 0)) = stK[q];
        }
        if (tid < m) sd[tid] = std_;
        // the trajectory stores of the previous step go out here, BEFORE the next prefetch is issued: memory
        // operations retire in order, so the wait for that prefetch at the top of the next step then never
        // includes a younger store's round trip (the stores are invisible to the compiler's wait counts)
        if (active) {
            store_vec(Xw0 + (unsigned)(xw_off + t * n), x, NS);
            if (t > 0) store_vec(Uw0 + uw_off_at(t - 1), ut, NC);
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
        wave_sync<NW>();
        if (t + PF < T) fetch(sg, t + PF);
        if (a == 0 && t > 0) stage_cost(J, (t & 1) ^ 1);   // stage cost of step t-1
        DPILQR_LDS_FENCE();
        // du = K[t] dx + alpha d[t] (control.py:106): this agent's NC rows, j ascending
        {
            double sum[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) sum[c] = 0.0;
            const double* rows = sK + a * (W::KBLK + W::KGAP);
            if (n % 2 == 0) {
#pragma unroll 5
                for (int j = 0; j < n; j += 2) {
                    const v2d dx2 = *reinterpret_cast<const v2d*>(lds + W::odx + g * W::LDG + j);
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
                    const double dxj = lds[W::odx + g * W::LDG + j];
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
        const double cr = ref_cost<NS, NC>(x, ut, xf, Q, R, false);
        double cp[PPL > 0 ? PPL : 1];
#pragma unroll
        for (int q = 0; q < PPL; ++q) cp[q] = pair_cost_nd<NS>(sxs + (pp[q] & 255) * NS, sxs + ((pp[q] >> 8) & 255) * NS, pp[q] >> 16, radius);
        DPILQR_LDS_FENCE();
        post_costs(cr, cp, t & 1);
        double xn[NS];
        integrate<MODEL>(x, ut, D.dt, xn);
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = xn[i];
        wave_sync<NW>();   // everybody is done with this step's staged K, d, dx, x'
    };
    if constexpr (PF == 1) {
        for (int t = 0; t < T; ++t) step(stg[0], t);
    } else {
        int t = 0;
        for (; t + 1 < T; t += 2) { step(stg[0], t); step(stg[1], t + 1); }
        if (t < T) step(stg[0], t);
    }
#ifdef DPILQR_PHASE_STAMPS
    if (g_stamp_buf && tid == 0) {
        g_stamp_buf[2 * slot] = w_wait;
        g_stamp_buf[2 * slot + 1] = __builtin_amdgcn_s_memtime() - w_t0;
    }
#endif
    if (active) {
        store_vec(Xw0 + (unsigned)(xw_off + T * n), x, NS);
        if (T > 0) store_vec(Uw0 + uw_off_at(T - 1), ut, NC);
    }
    {
        // last stage cost, then the terminal cost cost(X[T], 0, terminal=True) (control.py:112)
        if (active) {
#pragma unroll
            for (int i = 0; i < NS; ++i) sxs[a * NS + i] = x[i];
        }
        wave_sync<NW>();
        if (a == 0 && T > 0) stage_cost(J, (T & 1) ^ 1);
        if constexpr (NS >= 6 || NW > 1) {
            // the terminal step's pair table, made again from a lane id the optimiser cannot see through: the offsets the horizon
            // loop derives from it need not stay alive across the loop for this one use (they were spilled before it and reloaded here)
            int a2 = a;
            asm volatile("" : "+v"(a2));
            make_pairs(a2);
        }
        double Qf[NS * NS], uz[NC];
#pragma unroll
        for (int i = 0; i < NS * NS; ++i) Qf[i] = P.Qf[a * NS * NS + i];
#pragma unroll
        for (int c = 0; c < NC; ++c) uz[c] = 0.0;
        const double cr = ref_cost<NS, NC>(x, uz, xf, Qf, R, true);
        double cp[PPL > 0 ? PPL : 1];
#pragma unroll
        for (int q = 0; q < PPL; ++q) cp[q] = pair_cost_nd<NS>(sxs + (pp[q] & 255) * NS, sxs + ((pp[q] >> 8) & 255) * NS, pp[q] >> 16, radius);
        DPILQR_LDS_FENCE();
        post_costs(cr, cp, T & 1);
        wave_sync<NW>();
        if (a == 0) stage_cost(J, T & 1);
    }
    if constexpr (NS >= 6 || NW > 1) {      // (g from the lane id again: not a register kept, or spilled, across the horizon loop for this one store)
        int tid2 = tid;
        asm volatile("" : "+v"(tid2));
        if (active && a == 0) lds[W::oJ + tid2 / KA] = J;
    } else {
        if (active && a == 0) lds[W::oJ + g] = J;
    }
    // the candidates' trajectory stores (issued behind the compiler's back) must have landed before the copy below
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    wave_sync<NW>();
    int* ctl = reinterpret_cast<int*>(lds + W::octl);
    if (tid == 0) ctl[0] = linesearch_decide(S, b, NG, lds + W::oJ);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    wave_sync<NW>();
    const int acc = ctl[0];
    if (acc < 0) return;
    // accepted: X, U <- the accepted candidate's trajectory (a coalesced copy out of the scratch)
    const double* Xa = Xc + ((int64_t)slot * NG + acc) * (int64_t)(T + 1) * n;
    const double* Ua = Uc + ((int64_t)slot * NG + acc) * (int64_t)T * m;
    for (int e = tid; e < (T + 1) * n; e += NTH) Xb[e] = Xa[e];
    for (int e = tid; e < T * m; e += NTH) Ub[e] = Ua[e];
}

// ---- rollouts (ilqrSolver._rollout, control.py:80-93) for a batch of one model: 64 / KA sub-problems share a
// wavefront, lane (w, a) = (sub-problem w of the wavefront, agent a).  The generic kernel gives every sub-problem a
// wavefront of which KA lanes work; the solve starts with one rollout per item of the whole job.
template <int MODEL, int KA>
struct WaveRolloutLds {
    static constexpr int NS = ModelDef<MODEL>::NS, n = KA * NS, NP = KA * (KA - 1) / 2, NP1 = NP > 0 ? NP : 1;
    static constexpr int IPW = 64 / KA;                       // sub-problems per wavefront
    static constexpr int oxs = 0, ocr = oxs + IPW * n, ocp = ocr + IPW * KA, total = (ocp + IPW * NP1 + 1) & ~1;
};

template <int MODEL, int KA>
__global__ __launch_bounds__(256) void k_rollout_wave(dpilqr_batch_desc D, const double* __restrict__ x0,
                                                      const double* __restrict__ U, double* __restrict__ X,
                                                      double* __restrict__ Jout) {
    using W = WaveRolloutLds<MODEL, KA>;
    constexpr int NS = W::NS, NC = ModelDef<MODEL>::NC, n = W::n, m = KA * NC, NPAIRS = W::NP, NP1 = W::NP1, IPW = W::IPW;
    constexpr PairTable<KA> PT{};
    constexpr int PPL = (NPAIRS + KA - 1) / KA;
    const int sub = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tid = threadIdx.x & 63;
    const bool lane_on = tid < IPW * KA;
    const int w = lane_on ? tid / KA : 0, a = lane_on ? tid - (tid / KA) * KA : 0;
    const int64_t b_raw = ((int64_t)blockIdx.x * 4 + sub) * IPW + w;
    const bool active = lane_on && b_raw < D.B;
    const int b = (int)(b_raw < D.B ? b_raw : D.B - 1);       // idle lanes shadow the last item, store nothing
    const int T = D.T;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    double* lds = lds_all + sub * W::total;
    double* sxs = lds + W::oxs + w * n;
    double* scr = lds + W::ocr + w * KA;
    double* scp = lds + W::ocp + w * NP1;
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
    int pi[PPL > 0 ? PPL : 1], pj[PPL > 0 ? PPL : 1], pnd[PPL > 0 ? PPL : 1];
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const int p = min(a + q * KA, NP1 - 1);
        int ii = 0, jj = 0;
#pragma unroll
        for (int e = 0; e < NPAIRS; ++e)
            if (e == p) { ii = PT.i[e]; jj = PT.j[e]; }
        pi[q] = ii; pj[q] = jj;
        pnd[q] = homog ? 2 : min(P.n_dims[ii], P.n_dims[jj]);
    }
    const double* Ub = U + (int64_t)b * T * m + a * NC;
    double* Xw = X + (int64_t)b * (T + 1) * n + a * NS;
    auto store_vec = [&](double* p, const double* v, int len) {
        if ((len & 1) == 0) {
#pragma unroll
            for (int i = 0; i < len; i += 2) store_v2d_nt(p + i, v2d{v[i], v[i + 1]});
        } else {
#pragma unroll
            for (int i = 0; i < len; ++i) store_f64_nt(p + i, v[i]);
        }
    };
    auto stage_cost = [&](double& J) {   // pairs (combinations order), agents, then time: the reference's order
        double prox = 0.0, ref = 0.0;
#pragma unroll
        for (int p = 0; p < NPAIRS; ++p) prox += scp[p];
#pragma unroll
        for (int i = 0; i < KA; ++i) ref += scr[i];
        J += D.w_prox * prox + D.w_ref * ref;
    };
    double x[NS], u[NC], un[NC];
#pragma unroll
    for (int i = 0; i < NS; ++i) x[i] = x0[(int64_t)b * n + a * NS + i];
#pragma unroll
    for (int i = 0; i < NC; ++i) un[i] = (T > 0) ? Ub[i] : 0.0;
    double J = 0.0;
    for (int t = 0; t < T; ++t) {
        if (active) store_vec(Xw + (int64_t)t * n, x, NS);
#pragma unroll
        for (int i = 0; i < NC; ++i) u[i] = un[i];
        if (t + 1 < T) {
#pragma unroll
            for (int i = 0; i < NC; ++i) un[i] = Ub[(int64_t)(t + 1) * m + i];
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) sxs[a * NS + i] = x[i];
        DPILQR_LDS_FENCE();
        if (a == 0 && t > 0) stage_cost(J);
        const double cr = ref_cost<NS, NC>(x, u, xf, Q, R, false);
        double cp[PPL > 0 ? PPL : 1];
#pragma unroll
        for (int q = 0; q < PPL; ++q) cp[q] = pair_cost_nd<NS>(sxs + pi[q] * NS, sxs + pj[q] * NS, pnd[q], radius);
        DPILQR_LDS_FENCE();
        scr[a] = cr;
#pragma unroll
        for (int q = 0; q < PPL; ++q)
            if (a + q * KA < NPAIRS) scp[a + q * KA] = cp[q];
        double xn[NS];
        integrate<MODEL>(x, u, D.dt, xn);
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = xn[i];
    }
    if (active) store_vec(Xw + (int64_t)T * n, x, NS);
#pragma unroll
    for (int i = 0; i < NS; ++i) sxs[a * NS + i] = x[i];
    DPILQR_LDS_FENCE();
    if (a == 0 && T > 0) stage_cost(J);
    {
        double Qf[NS * NS], uz[NC];
#pragma unroll
        for (int i = 0; i < NS * NS; ++i) Qf[i] = P.Qf[a * NS * NS + i];
#pragma unroll
        for (int c = 0; c < NC; ++c) uz[c] = 0.0;
        const double cr = ref_cost<NS, NC>(x, uz, xf, Qf, R, true);
        double cp[PPL > 0 ? PPL : 1];
#pragma unroll
        for (int q = 0; q < PPL; ++q) cp[q] = pair_cost_nd<NS>(sxs + pi[q] * NS, sxs + pj[q] * NS, pnd[q], radius);
        DPILQR_LDS_FENCE();
        scr[a] = cr;
#pragma unroll
        for (int q = 0; q < PPL; ++q)
            if (a + q * KA < NPAIRS) scp[a + q * KA] = cp[q];
        DPILQR_LDS_FENCE();
        if (a == 0) stage_cost(J);
    }
    if (active && a == 0) Jout[b] = J;
}

}  // namespace dpilqr
