// forward.hpp -- K3: rollouts, the line-searched forward pass and the accept / regularisation logic.
//
// Reference: ilqrSolver._rollout (control.py:80-93), _forward_pass (:95-114) and the body of the
// iteration loop in solve (:179-211) with _decrease_regularization (:232-237).
//
// One workgroup per sub-problem; thread (g, a) = (line-search candidate g, agent a).  Every candidate
// alpha of the reference's sequential search is independent given (X,U,K,d), so all are rolled out at
// once and the FIRST one (in table order) with J < J* is accepted -- identical to the sequential
// search, including the count of forward passes it would have made.  The accepted candidate is then
// rolled out once more, writing X,U in place (no per-candidate trajectory scratch in HBM).
// Per time step the agents of a candidate exchange dx and positions through LDS; the stage cost is
// summed in the reference's order (pairs in combinations order, agents in order, then over time).
#pragma once
#include <hip/hip_runtime.h>

#include "cost.hpp"
#include "models.hpp"
#include "solve_state.hpp"

namespace dpilqr {

// 16x16x4 matrix-pipe tile (the same helper as riccati_big.hpp's Mfma<R>; this header is compiled in other translation units): lane
// (g = lane / 16, c = lane % 16) supplies A[row c][reduction g], B[reduction g][column c] and owns rows row(v, g) of column c of D
template <typename R> struct FwdMfma;
template <> struct FwdMfma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    __device__ static __forceinline__ acc_t mac(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int v, int g) { return g + 4 * v; }
};
template <> struct FwdMfma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    __device__ static __forceinline__ acc_t mac(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int v, int g) { return 4 * g + v; }
};

constexpr int kFwdBatch = 12;  // PIPE: reduction steps of K[t] dx per register set of A operands, two sets (forward.hpp, horizon_pass)
constexpr int kMaxStage = 16;  // K[t] elements a thread stages per step (per chunk, kdirect): ceil(n_u*n_x / threads) must not exceed this

struct ForwardLds {   // offsets in elements of the arithmetic type
    int Kt, dt, dx, xs, cref, cpair, J, ctl, du, total;
    int cw, rs;   // kdirect: columns of K[t] per chunk, elements per chunk column (>= m)
    int mt;       // kdirect: m rounded up to the matrix pipe's 16-row tiles (the row stride of `du`)
    // kdirect: K[t] goes through LDS in CHUNKS of cw columns (large clusters: K[t] alone is 154 KB at n_x = 240, n_u = 80), two
    // chunk buffers used in turn, a chunk stored column by column with row a NC + c of a column at c k + a (the lanes of a candidate --
    // its agents -- read consecutive words, the candidates the same ones: with an agent's NC rows contiguous the reads were 5-way
    // bank conflicts).  Round 5; before, every thread walked its rows in global memory, two columns in flight: 160 k of a step's
    // 265 k clocks at cfg5's size (now 68 k, bound by the LDS return path: every entry of K[t] is read once per candidate).
    __host__ __device__ ForwardLds(int n, int m, int k, int ngrp, bool kdirect = false) {
        const int npairs = k * (k - 1) / 2;
        int o = 0;
        cw = 0; rs = 0;
        if (kdirect) {
            const int nth = ((k * ngrp + 63) / 64) * 64;                  // the launch's threads (tu_bigfwd.hip)
            int c = 2048 / m;                                             // two buffers of <= 2048 elements
            if (c > kMaxStage * nth / m) c = kMaxStage * nth / m;         // ... which the threads stage kMaxStage elements each
            if (c > n) c = n;
            cw = c < 1 ? 1 : c;
            rs = ((m + 14) / 16) * 16 + 1;   // >= m and = 1 mod 16: the staging lanes' consecutive columns fall into consecutive banks
        }
        // everything staged per time step is double-buffered by the parity of t: one barrier per step
        Kt = o;    o += kdirect ? 2 * (cw * rs + 2) : 2 * m * n;   // (kdirect: + the store target of idle elements)
        dt = o;    o += 2 * m;
        dx = o;    o += 2 * ngrp * n;
        xs = o;    o += 2 * ngrp * n;
        cref = o;  o += 2 * ngrp * k;
        cpair = o; o += 2 * ngrp * (npairs > 0 ? npairs : 1);
        J = o;     o += ngrp;
        ctl = o;   o += 2;
        o = (o + 1) & ~1;
        mt = ((m + 15) / 16) * 16;
        du = o;    o += kdirect ? ngrp * mt : 0;      // kdirect: K[t] dx of every candidate, from the matrix pipe's tiles to the agents' lanes
        total = (o + 1) & ~1;
    }
};
inline size_t forward_lds_bytes(int n, int m, int k, int ngrp, bool kdirect = false, size_t elem = sizeof(double)) {
    return elem * (size_t)ForwardLds(n, m, k, ngrp, kdirect).total;
}

// Workgroup-wide LDS hand-off.  A single-wave workgroup executes its LDS operations in order, so a
// compiler fence is enough; larger workgroups use a bare s_barrier behind an LDS-only wait (NOT
// __syncthreads(), whose vmcnt(0) would drain the global prefetches that are meant to stay in flight).
__device__ __forceinline__ void lds_handoff(bool single_wave) {
    if (single_wave) asm volatile("" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// p[0] + p[1] + ... + p[count - 1] added in that order (the reference's stage-cost sums).  One thread per candidate walks 190 pair
// costs at cfg5's size while the rest of its wavefront waits: the loads sixteen at a time and a set AHEAD of the additions (a
// load -> add -> load chain is an LDS round trip per term; eight at a time without the look-ahead: 7 k of a step's 135 k clocks)
template <typename R>
__device__ __forceinline__ R sum_in_order(const R* p, int count) {
    constexpr int W = 16;
    R s = 0.0;
    int i = 0;
    if (count >= 2 * W) {
        R va[W], vb[W];
#pragma unroll
        for (int q = 0; q < W; ++q) va[q] = p[q];
        for (; i + 2 * W <= count; i += 2 * W) {
#pragma unroll
            for (int q = 0; q < W; ++q) vb[q] = p[i + W + q];
#pragma unroll
            for (int q = 0; q < W; ++q) s += va[q];
            // (the set after next; beyond the end it re-reads the array's last full set -- never used)
            const int nx = i + 3 * W <= count ? i + 2 * W : count - W;
#pragma unroll
            for (int q = 0; q < W; ++q) va[q] = p[nx + q];
#pragma unroll
            for (int q = 0; q < W; ++q) s += vb[q];
        }
        if (i + W <= count) {      // va holds p[i .. i + W)
#pragma unroll
            for (int q = 0; q < W; ++q) s += va[q];
            i += W;
        }
    }
    for (; i + 8 <= count; i += 8) {
        R v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = p[i + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; i < count; ++i) s += p[i];
    return s;
}

// One pass over the horizon for the calling thread's (candidate g, agent a).
//   GAINS : u = U + (K dx + alpha d) (control.py:104-107) ; else u = U (control.py:89)
//   Xw/Uw : where to write this candidate's trajectory (null = nowhere); never aliases Xold/Uold.
//   R     : arithmetic type (double; float in the fp32 arm of BASELINE config 5's tolerance study)
// Everything a step needs from HBM (K[t], d[t], X[t], U[t]) is fetched one step ahead into registers.
// Returns J on the a == 0 lane of each candidate.
template <typename R, int NS, int NC, bool GAINS, bool KDIRECT, bool PIPE = false>
__device__ R horizon_pass(const dpilqr_batch_desc& D, const ItemParams& P, bool homog, int tid, int nth, bool active,
                          int g, int a, int ngrp, const R* x_init, const R* __restrict__ Xold,
                          const R* __restrict__ Uold, const R* __restrict__ Kb,
                          const R* __restrict__ db, R alpha, R* __restrict__ Xw,
                          R* __restrict__ Uw, R* lds) {
    const int k = D.k, T = D.T, n = k * NS, m = k * NC;
    const int npairs = k * (k - 1) / 2, np1 = npairs > 0 ? npairs : 1;
    const ForwardLds O(n, m, k, ngrp, KDIRECT);
    const int gg = active ? g : 0;
    const bool single_wave = nth <= 64;
    const int mn = m * n;
    const R dtr = (R)D.dt, radius = (R)P.radius, w_prox = (R)D.w_prox, w_ref = (R)D.w_ref;

    R x[NS], xold[NS], u[NC], stK[kMaxStage], std_ = 0.0;
    const int model = active ? P.model[a] : 0;
    const double* xf = P.xf + a * NS;
    const double* Qa = P.Q + a * NS * NS;
    const double* Ra = P.R + a * NC * NC;
    const double* Qfa = P.Qf + a * NS * NS;
    // PIPE (round 6): the running cost's weights and the goals staged once in LDS, in the arithmetic type, where the chunk buffers of
    // the other form would be: every lane read its agent's Q -- 144 entries at twelve states -- from global memory at every step, one
    // load instruction per entry with twenty distinct lines in it, and the four wavefronts' 576 such instructions per step were 13 k of
    // a step's 135 k clocks at cfg5's size (the one-wavefront rollout: 3 k).  Rows of an agent 1 mod 2 doubles apart from the next
    // agent's: the candidates of an agent read one word, the agents of a candidate different banks.
    constexpr int kWq = NS * NS + 1, kWr = NC * NC, kWx = NS;
    constexpr bool w_lds = PIPE && KDIRECT && GAINS;      // (the launcher's promise that they fit: forward_on_pipe)
    R* const sWq = lds + O.Kt;
    R* const sWr = sWq + k * kWq;
    R* const sWx = sWr + k * kWr;
    if (w_lds) {
        for (int e = tid; e < k * NS * NS; e += nth) sWq[(e / (NS * NS)) * kWq + e % (NS * NS)] = (R)P.Q[e];
        for (int e = tid; e < k * kWr; e += nth) sWr[e] = (R)P.R[e];
        for (int e = tid; e < k * kWx; e += nth) sWx[e] = (R)P.xf[e];
        lds_handoff(single_wave);
    }

    // KDIRECT: K[t] in chunks of O.cw columns.  Element e = tid + q nth of a chunk is (row e / cw, column e % cw): consecutive threads
    // read consecutive columns of a row (coalesced) and store column by column (ForwardLds)
    const int n_chunks = KDIRECT ? (n + O.cw - 1) / O.cw : 0;
    int ck_src[KDIRECT ? kMaxStage : 1], ck_dst[KDIRECT ? kMaxStage : 1];
    if constexpr (KDIRECT && GAINS && !PIPE) {
#pragma unroll
        for (int q = 0; q < kMaxStage; ++q) {
            const int e = tid + q * nth;
            const int row = e / O.cw, jj = e - row * O.cw;
            ck_src[q] = (e < m * O.cw) ? row * n + jj : -1;
            ck_dst[q] = jj * O.rs + (row % NC) * k + row / NC;
        }
    }
    auto fetch_chunk = [&](int t, int ch) {  // registers <- HBM for chunk ch of K[t]
        if constexpr (KDIRECT && GAINS && !PIPE) {
            const R* Kt = Kb + (int64_t)t * mn + ch * O.cw;
            const int left = n - ch * O.cw;   // columns of the matrix from this chunk's first on
            // (no load behind a test: an element outside the chunk -- e >= m cw, or column jj = dst / rs >= left in the last chunk --
            // reads the chunk's first entry instead; behind per-lane tests the compiler drained the load queue before every load)
#pragma unroll
            for (int q = 0; q < kMaxStage; ++q) stK[q] = Kt[(ck_src[q] >= 0 && ck_dst[q] < left * O.rs) ? ck_src[q] : 0];
        }
    };
    auto fetch = [&](int t) {  // registers <- HBM for step t
        if (GAINS) {
            if constexpr (!KDIRECT) {
                const R* Kt = Kb + (int64_t)t * mn;
#pragma unroll
                for (int q = 0; q < kMaxStage; ++q) {
                    const int e = tid + q * nth;
                    if (e < mn) stK[q] = Kt[e];
                }
            }
            if (tid < m) std_ = db[(int64_t)t * m + tid];
        }
        if (active) {
#pragma unroll
            for (int i = 0; i < NC; ++i) u[i] = Uold[(int64_t)t * m + a * NC + i];
            if (GAINS) {
#pragma unroll
                for (int i = 0; i < NS; ++i) xold[i] = Xold[(int64_t)t * n + a * NS + i];
            }
        }
    };

    // the dimensions of this agent's pairs (min of the two agents' n_dims, cost.py:145), two bits per partner offset: read from the
    // descriptor once -- inside the pair loop the two loads per pair and step were most of its 10 k clocks at cfg5's size
    unsigned long long nd_pack = 0ull;
    if (active && !homog) {
        for (int dd = 1; 2 * dd <= k && dd < 32; ++dd) {
            const int o = a + dd < k ? a + dd : a + dd - k;
            nd_pack |= (unsigned long long)(min(P.n_dims[a], P.n_dims[o]) & 3) << (2 * dd);
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = x_init[a * NS + i];
        if (Xw) {
#pragma unroll
            for (int i = 0; i < NS; ++i) Xw[a * NS + i] = x[i];
        }
    }
    // PIPE: this lane's A operands (see the products below), kFwdBatch reduction steps per set, two sets used in turn
    R pkA0[PIPE ? kFwdBatch : 1], pkA1[PIPE ? kFwdBatch : 1], pkB0[PIPE ? kFwdBatch : 1], pkB1[PIPE ? kFwdBatch : 1];
    bool pk_two = false;
    int64_t pk_o0 = 0, pk_o1 = 0;
    if constexpr (PIPE && KDIRECT && GAINS) {
        const int wv = tid >> 6, ln = tid & 63, g16 = ln >> 4, c16 = ln & 15, nw = nth >> 6;
        const int r0 = 16 * wv + c16, r1 = 16 * (wv + nw) + c16;
        const bool pk_v0 = r0 < m, pk_v1 = r1 < m;
        pk_two = __builtin_amdgcn_readfirstlane((int)(16 * (wv + nw) < O.mt)) != 0;    // this wavefront has a second row tile
        pk_o0 = (int64_t)(pk_v0 ? r0 : 0) * n; pk_o1 = (int64_t)(pk_v1 ? r1 : 0) * n;
        (void)g16;
    }
    fetch(0);
    fetch_chunk(0, 0);
    int ck_buf = 0;   // KDIRECT: the chunk buffer the next chunk goes into
    R J = 0.0;

#ifdef DPILQR_FWD_STAMPS      // diagnostic builds only: thread 0's clock per part of a step (scripts/r06_fwd_phases.sh)
    unsigned long long fph[6] = {0, 0, 0, 0, 0, 0}, fph_t = __builtin_amdgcn_s_memtime();
#define FPHASE(i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); fph[i] += now_ - fph_t; fph_t = now_; }
#else
#define FPHASE(i)
#endif
    for (int t = 0; t < T; ++t) {
        const int par = t & 1;
        R* sKt = lds + O.Kt + par * (KDIRECT ? 0 : mn);
        R* sdt = lds + O.dt + par * m;
        R* sdx = lds + O.dx + (par * ngrp + gg) * n;
        R* sxs = lds + O.xs + (par * ngrp + gg) * n;
        R ut[NC];
        if (GAINS) {
            if constexpr (!KDIRECT) {
#pragma unroll
                for (int q = 0; q < kMaxStage; ++q) {
                    const int e = tid + q * nth;
                    if (e < mn) sKt[e] = stK[q];
                }
            }
            if (tid < m) sdt[tid] = std_;
        }
        if (active) {
#pragma unroll
            for (int i = 0; i < NC; ++i) ut[i] = u[i];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                if (GAINS) sdx[a * NS + i] = x[i] - xold[i];  // dx = X'[t] - X[t]
                sxs[a * NS + i] = x[i];
            }
        }
        if (t + 1 < T) fetch(t + 1);
        lds_handoff(single_wave);
        FPHASE(0)
        R ksum[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) ksum[c] = 0.0;
        if constexpr (KDIRECT && GAINS) {
            // K[t] dx of ALL candidates at once.  PIPE (round 6): du[row][candidate] += K[row][j] dx_candidate[j] is a (m x n)(n x
            // n_alpha) product on the MATRIX PIPE, 16 x 16 x 4 tiles with the candidates as the tile's columns (10 of 16), the row tiles
            // dealt to the wavefronts (forward_on_pipe: at most two each) -- j ascending, one multiply-add per term.  Otherwise
            // (rounds 2-5) chunk by chunk through LDS: this chunk from the registers into its buffer, the next one requested (the first
            // of step t + 1 behind the last of step t), one barrier, then every (candidate, agent) lane walks the chunk's columns --
            // every entry of K[t] read from LDS once per candidate, 68 k of a step's 170 k clocks at cfg5's size.  The buffer written
            // in one round was last read two rounds earlier, and every thread has passed the barrier in between.
            if constexpr (PIPE) {
                // Round 6, second form: the matrix pipe's A operand straight from global memory.  Lane (g16, c16) of a row tile supplies
                // K[t][16 it + c16][4 s + g16] in reduction step s -- its own row, every fourth column -- so a wavefront's load is sixteen
                // 32-byte pieces of sixteen rows, four consecutive steps share a cache line, and nothing is staged: no chunk buffers, no
                // barrier per chunk, no round of global latency per chunk (ten chunks per step, each one load in flight deep, were 45 k
                // of a step's 160 k clocks at cfg5's size; scripts/r06_fwd_phases.sh).  Two register sets of kFwdBatch reduction steps
                // each are requested before the first product and a set is requested again as soon as its products are issued.
                // (Measured and dropped: touching K[t + 1]'s cache lines a step ahead so that they wait in the L2 -- no change.)
                // The wavefront's own instruction stream matters as much (one wavefront per SIMD issues an instruction every four
                // clocks at best; a first version spent ~40 instructions per reduction step on addresses, clamps and selects): a step
                // is ONE 64-bit address per row tile and batch, every load and every LDS read of the batch at an immediate offset from
                // it, and no select at all -- rows of K beyond n_u and candidates beyond n_alpha produce rows / columns of the tile
                // that nobody reads (a product's rows and columns do not mix), so their operands are whatever the clamped addresses
                // hold.  Only a batch that reaches past column n_x (none at cfg5's size) takes the form with clamps and zeroed
                // operands.  Same products, same order: j ascending, one multiply-add per term.  The phase: 45 k -> 25 k clocks per step
                // (of which 14 k without any operand load: 120 dependent products of 64 clocks on the wavefront that holds two row
                // tiles, the LDS reads, the hand-over of du); the ten-candidate pass 9.96 -> 8.3 ms.
                typedef typename FwdMfma<R>::acc_t acc_t;
                const int wv = tid >> 6, ln = tid & 63, g16 = ln >> 4, c16 = ln & 15, nw = nth >> 6;
                const int tiles = O.mt / 16;
                acc_t acc0 = acc_t{0, 0, 0, 0}, acc1 = acc_t{0, 0, 0, 0};
                const int it0 = wv, it1 = wv + nw;
                const bool two = pk_two;
                const bool bv = c16 < ngrp;
                const R* dxc = lds + O.dx + (par * ngrp + (bv ? c16 : 0)) * n;
                const int nks = (n + 3) >> 2;
                const R* Kt = Kb + (int64_t)t * mn;
                auto whole = [&](int s0) { return 4 * (s0 + kFwdBatch) <= n; };     // every column of the batch exists, in every lane group
                auto load = [&](R (&k0)[kFwdBatch], R (&k1)[kFwdBatch], int s0) __attribute__((always_inline)) {
                    if (whole(s0)) {
                        const R* q0 = Kt + pk_o0 + 4 * s0 + g16;
#pragma unroll
                        for (int q = 0; q < kFwdBatch; ++q) k0[q] = q0[4 * q];
                        if (two) {      // (wave-uniform, and known to be: a branch, not a masked region around every load)
                            const R* q1 = Kt + pk_o1 + 4 * s0 + g16;
#pragma unroll
                            for (int q = 0; q < kFwdBatch; ++q) k1[q] = q1[4 * q];
                        }
                    } else {        // (no load behind a test: a column beyond the row's end reads the row's last entry instead)
#pragma unroll
                        for (int q = 0; q < kFwdBatch; ++q) k0[q] = Kt[pk_o0 + min(4 * (s0 + q) + g16, n - 1)];
                        if (two) {
#pragma unroll
                            for (int q = 0; q < kFwdBatch; ++q) k1[q] = Kt[pk_o1 + min(4 * (s0 + q) + g16, n - 1)];
                        }
                    }
                };
                auto products = [&](const R (&k0)[kFwdBatch], const R (&k1)[kFwdBatch], int s0) __attribute__((always_inline)) {
                    R bq[kFwdBatch];
                    if (whole(s0)) {
                        const R* d0 = dxc + 4 * s0 + g16;
#pragma unroll
                        for (int q = 0; q < kFwdBatch; ++q) bq[q] = d0[4 * q];
#pragma unroll
                        for (int q = 0; q < kFwdBatch; ++q) acc0 = FwdMfma<R>::mac(k0[q], bq[q], acc0);
                        if (two) {
#pragma unroll
                            for (int q = 0; q < kFwdBatch; ++q) acc1 = FwdMfma<R>::mac(k1[q], bq[q], acc1);
                        }
                    } else {        // a column beyond the last: both operands zero, the product adds nothing
                        bool jv[kFwdBatch];
#pragma unroll
                        for (int q = 0; q < kFwdBatch; ++q) {
                            const int jj = 4 * (s0 + q) + g16;
                            jv[q] = jj < n;
                            R b = dxc[jv[q] ? jj : 0];
                            asm volatile("" : "+v"(b));      // (requested whatever jv says: behind the test, every read is a masked region with its own wait)
                            bq[q] = jv[q] ? b : (R)0.0;
                        }
#pragma unroll
                        for (int q = 0; q < kFwdBatch; ++q) acc0 = FwdMfma<R>::mac(jv[q] ? k0[q] : (R)0.0, bq[q], acc0);
                        if (two) {
#pragma unroll
                            for (int q = 0; q < kFwdBatch; ++q) acc1 = FwdMfma<R>::mac(jv[q] ? k1[q] : (R)0.0, bq[q], acc1);
                        }
                    }
                };
                load(pkA0, pkA1, 0);
                load(pkB0, pkB1, kFwdBatch);
                for (int s0 = 0; s0 < nks; s0 += 2 * kFwdBatch) {
                    products(pkA0, pkA1, s0);
                    if (s0 + 2 * kFwdBatch < nks) load(pkA0, pkA1, s0 + 2 * kFwdBatch);
                    if (s0 + kFwdBatch < nks) products(pkB0, pkB1, s0 + kFwdBatch);
                    if (s0 + 3 * kFwdBatch < nks) load(pkB0, pkB1, s0 + 3 * kFwdBatch);
                }
                // the tiles' entries to where the agents' lanes find them: du[candidate][row]
                R* sdu = lds + O.du;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int ra = 16 * it0 + FwdMfma<R>::row(v, g16), rb = 16 * it1 + FwdMfma<R>::row(v, g16);
                    if (bv && it0 < tiles) sdu[c16 * O.mt + ra] = acc0[v];
                    if (bv && two) sdu[c16 * O.mt + rb] = acc1[v];
                }
                lds_handoff(single_wave);
                if (active) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) ksum[c] = sdu[g * O.mt + a * NC + c];
                }
            } else {
            for (int ch = 0; ch < n_chunks; ++ch) {
                R* sKc = lds + O.Kt + ck_buf * (O.cw * O.rs + 2);
                const int left = n - ch * O.cw, cwa = left < O.cw ? left : O.cw;
#pragma unroll
                for (int q = 0; q < kMaxStage; ++q)
                    sKc[(ck_src[q] >= 0 && ck_dst[q] < left * O.rs) ? ck_dst[q] : O.cw * O.rs + (tid & 1)] = stK[q];   // (else: the pad behind the buffer)
                if (ch + 1 < n_chunks) fetch_chunk(t, ch + 1);
                else if (t + 1 < T) fetch_chunk(t + 1, 0);
                lds_handoff(single_wave);
                  if (active) {
                    const R* colp = sKc + a;
                    const R* dxp = sdx + ch * O.cw;
                    // eight columns' operands requested before the first is used (one wavefront per SIMD: nothing else hides an
                    // LDS round trip), then added in ascending order
                    int jj = 0;
                    for (; jj + 8 <= cwa; jj += 8) {
                        R kv[8][NC], dxv[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            dxv[q] = dxp[jj + q];
#pragma unroll
                            for (int c = 0; c < NC; ++c) kv[q][c] = colp[(jj + q) * O.rs + c * k];
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int c = 0; c < NC; ++c) ksum[c] += kv[q][c] * dxv[q];
                    }
                    for (; jj < cwa; ++jj) {
                        const R dxj = dxp[jj];
#pragma unroll
                        for (int c = 0; c < NC; ++c) ksum[c] += colp[jj * O.rs + c * k] * dxj;
                    }
                  }
                ck_buf ^= 1;
            }
            }
        }
        FPHASE(1)
        if (active && a == 0 && t > 0) {  // stage cost of step t-1 (other parity), summed in the reference's order
            const R* cr = lds + O.cref + ((par ^ 1) * ngrp + g) * k;
            const R* cp = lds + O.cpair + ((par ^ 1) * ngrp + g) * np1;
            const R prox = sum_in_order(cp, npairs), ref = sum_in_order(cr, k);
            J += w_prox * prox + w_ref * ref;
        }
        FPHASE(2)
        if (active) {
            if (GAINS) {  // du = K[t] dx + alpha d[t] (control.py:106), this agent's NC rows, j ascending
                R sum[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) sum[c] = KDIRECT ? ksum[c] : (R)0.0;   // (KDIRECT: summed chunk by chunk above, j ascending)
                // the agent's NC rows of K[t], staged in LDS
                const R* rows = sKt + (a * NC) * n;
                if constexpr (KDIRECT) {
                } else if ((n & 1) == 0) {
                    typedef R v2r __attribute__((ext_vector_type(2)));
#pragma unroll 2
                    for (int j = 0; j < n; j += 2) {
                        const v2r dx2 = *reinterpret_cast<const v2r*>(sdx + j);
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const v2r kr = *reinterpret_cast<const v2r*>(rows + c * n + j);
                            sum[c] += kr.x * dx2.x;
                            sum[c] += kr.y * dx2.y;
                        }
                    }
                } else {
                    for (int j = 0; j < n; ++j) {
                        const R dxj = sdx[j];
#pragma unroll
                        for (int c = 0; c < NC; ++c) sum[c] += rows[c * n + j] * dxj;
                    }
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const R du = sum[c] + alpha * sdt[a * NC + c];
                    ut[c] = ut[c] + du;
                }
            }
            if constexpr (w_lds) lds[O.cref + (par * ngrp + g) * k + a] = ref_cost<NS, NC, R, R>(x, ut, sWx + a * kWx, sWq + a * kWq, sWr + a * kWr, false);
            else lds[O.cref + (par * ngrp + g) * k + a] = ref_cost<NS, NC>(x, ut, xf, Qa, Ra, false);
            FPHASE(3)
            // the candidate's pairs dealt evenly: agent a takes (a, a + 1), ..., (a, a + k / 2) mod k -- at most k / 2 each instead of
            // k - 1 for agent 0 -- each computed as (lower, higher) and put where the sum in combinations order finds it
            // (measured and dropped, round 6: four pairs at a time -- positions requested together, one test for the square root per group;
            // the loop got no faster, 9.1 k clocks per step, and the other phases' register shuffling cost more than it saved)
            for (int dd = 1; 2 * dd <= k; ++dd) {
                if (2 * dd == k && a >= dd) break;
                const int o = a + dd < k ? a + dd : a + dd - k;
                const int lo = a < o ? a : o, hi = a < o ? o : a;
                const int nd = homog ? 2 : (dd < 32 ? (int)((nd_pack >> (2 * dd)) & 3ull) : min(P.n_dims[lo], P.n_dims[hi]));
                lds[O.cpair + (par * ngrp + g) * np1 + pair_index(lo, hi, k)] =
                    pair_cost(sxs + lo * NS, sxs + hi * NS, nd, radius);
            }
            FPHASE(4)
            if (Uw) {
#pragma unroll
                for (int c = 0; c < NC; ++c) Uw[(int64_t)t * m + a * NC + c] = ut[c];
            }
            R xn[NS];
            integrate_rt<NS>(model, x, ut, dtr, xn);
#pragma unroll
            for (int i = 0; i < NS; ++i) x[i] = xn[i];
            if (Xw) {
#pragma unroll
                for (int i = 0; i < NS; ++i) Xw[(int64_t)(t + 1) * n + a * NS + i] = x[i];
            }
        }
        FPHASE(5)
    }
#ifdef DPILQR_FWD_STAMPS
    if (tid == 0 && blockIdx.x == 0)
        printf("horizon_pass<NS %d, GAINS %d> phases (thread 0's shader clocks per step): stage+handoff %.0f  K dx %.0f  stage-cost sum %.0f  du + ref_cost %.0f  pair costs %.0f  integrate + stores %.0f\n",
               NS, (int)GAINS, (double)fph[0] / T, (double)fph[1] / T, (double)fph[2] / T, (double)fph[3] / T, (double)fph[4] / T, (double)fph[5] / T);
#endif
    {
        // last stage cost, then the terminal cost cost(X[T], 0, terminal=True) (control.py:91,112)
        const int par = T & 1;
        R* sxs = lds + O.xs + (par * ngrp + gg) * n;
        if (active) {
#pragma unroll
            for (int i = 0; i < NS; ++i) sxs[a * NS + i] = x[i];
        }
        lds_handoff(single_wave);
        if (active) {
            if (a == 0 && T > 0) {
                const R* cr = lds + O.cref + ((par ^ 1) * ngrp + g) * k;
                const R* cp = lds + O.cpair + ((par ^ 1) * ngrp + g) * np1;
                const R prox = sum_in_order(cp, npairs), ref = sum_in_order(cr, k);
                J += w_prox * prox + w_ref * ref;
            }
            R uz[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) uz[c] = 0.0;
            lds[O.cref + (par * ngrp + g) * k + a] = ref_cost<NS, NC>(x, uz, xf, Qfa, Ra, true);
            // the candidate's pairs dealt evenly: agent a takes (a, a + 1), ..., (a, a + k / 2) mod k -- at most k / 2 each instead of
            // k - 1 for agent 0 -- each computed as (lower, higher) and put where the sum in combinations order finds it
            for (int dd = 1; 2 * dd <= k; ++dd) {
                if (2 * dd == k && a >= dd) break;
                const int o = a + dd < k ? a + dd : a + dd - k;
                const int lo = a < o ? a : o, hi = a < o ? o : a;
                const int nd = homog ? 2 : (dd < 32 ? (int)((nd_pack >> (2 * dd)) & 3ull) : min(P.n_dims[lo], P.n_dims[hi]));
                lds[O.cpair + (par * ngrp + g) * np1 + pair_index(lo, hi, k)] =
                    pair_cost(sxs + lo * NS, sxs + hi * NS, nd, radius);
            }
        }
        lds_handoff(single_wave);
        if (active && a == 0) {
            const R* cr = lds + O.cref + (par * ngrp + g) * k;
            const R* cp = lds + O.cpair + (par * ngrp + g) * np1;
            const R prox = sum_in_order(cp, npairs), ref = sum_in_order(cr, k);
            J += w_prox * prox + w_ref * ref;
        }
    }
    lds_handoff(single_wave);
    return J;
}

// kModeRollout    : X, J <- rollout(x0, U)                                  (control.py:80-93)
// kModeCandidates : Xc, Uc, Jc <- forward_pass(X, U, K, d, alpha_g) for every g (control.py:95-114)
// kModeLineSearch : the same, followed by the accept / regularisation logic of one solver iteration
//                   (control.py:179-211) and the copy of the accepted candidate into X, U.
// Workgroup layout: `ipb` sub-problems per workgroup.  When a sub-problem's threads fit one wavefront
// (n_alpha * k <= 64, e.g. cfg2's 50) four of them share a 256-thread workgroup, one wave each with its own
// LDS slice and no workgroup barrier -- the same SIMD-placement argument as for the sweep (riccati_tiled.hpp).
// R: arithmetic type; KDIRECT: K[t] is not staged in LDS (large clusters); lds_per_item in elements of R.
// the launcher's test for the matrix-pipe form of K[t] dx (KDIRECT, candidates / line-search mode): row tiles of 16 controls, at most
// two per wavefront; the candidates are a tile's columns
// ... and the running cost's weights and goals fit the LDS the chunk buffers of the other form would take (horizon_pass: sWq)
inline bool forward_on_pipe(int n, int m, int k, int threads, int ngrp) {
    const ForwardLds O(n, m, k, ngrp, true);
    const int ns = n / k, nc = m / k;
    return ((m + 15) / 16) <= 2 * (threads / 64) && ngrp <= 16 && 2 * (O.cw * O.rs + 2) >= k * (ns * ns + 1 + nc * nc + ns);
}

template <typename R, int NS, int NC, bool KDIRECT, bool PIPE = false>
__global__ __launch_bounds__(256, (KDIRECT || NS >= 12) ? 1 : 2) void k_forward(dpilqr_batch_desc D, int mode, const R* __restrict__ x0, R* X,
                                                  R* U, const R* __restrict__ K, const R* __restrict__ d,
                                                  const double* __restrict__ alphas, int ngrp, R* Xc, R* Uc,
                                                  double* Jc, SolveState S, const int32_t* __restrict__ items,
                                                  const int32_t* __restrict__ n_items, int ipb, int lds_per_item) {
    const int nth = (ipb > 1) ? 64 : (int)blockDim.x;
    const int sub = (ipb > 1) ? (int)(threadIdx.x >> 6) : 0;
    const int tid = (ipb > 1) ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const int slot = blockIdx.x * ipb + sub;
    if (slot >= (n_items ? *n_items : D.B)) return;
    const int b = items ? items[slot] : slot;
    const int k = D.k, T = D.T, n = k * NS, m = k * NC;
    int g = tid / k, a = tid - g * k;
    const ItemParams P = item_params(D, b);
    if constexpr (KDIRECT) {
        // Heterogeneous clusters (config 5: fourteen twelve-state quadcopters + six padded humans), candidates / line-search mode: a
        // wavefront that holds agents of BOTH models runs both models' integration one after the other, every step (the lanes of one
        // take no part in the other's).  Lanes dealt by CLASS instead -- the agents that share agent 0's model first, candidate by
        // candidate, padded to a wavefront boundary, then the others -- make every wavefront homogeneous where the padded layout
        // fits the workgroup (cfg5: 140 lanes in three wavefronts, 60 in the fourth).  Only which lane computes which (candidate,
        // agent) changes: same arithmetic, same results.  Round 6; the rollout's 64 threads and clusters of one model keep the plain deal.
        const int k0 = P.model[0];
        int nA = 0;
        for (int i = 0; i < k; ++i) nA += P.model[i] == k0;
        const int groups_ = mode == kModeRollout ? 1 : ngrp;
        const int lanesA = groups_ * nA, lanesB = groups_ * (k - nA), baseB = ((lanesA + 63) / 64) * 64;
        if (ipb == 1 && nA < k && baseB + lanesB <= nth) {
            const bool inB = tid >= baseB;
            const int idx = inB ? tid - baseB : tid, nC = inB ? k - nA : nA;
            const bool on = idx < (inB ? lanesB : lanesA);
            const int gi = idx / nC, ai = idx - gi * nC;
            int cnt = 0, ag = 0;      // the ai-th agent of the lane's class
            for (int i = 0; i < k; ++i) {
                const bool mine = (P.model[i] == k0) != inB;
                if (mine && cnt == ai) ag = i;
                cnt += mine;
            }
            g = on ? gi : groups_;      // (a lane of the padding: no candidate)
            a = on ? ag : 0;
        }
    }
    const bool homog = homogeneous_ndims(P.n_dims, k);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    R* lds = reinterpret_cast<R*>(lds_raw) + (size_t)sub * lds_per_item;
    const bool single_wave = nth <= 64;
    const ForwardLds O(n, m, k, ngrp, KDIRECT);
    R* Xb = X + (int64_t)b * (T + 1) * n;
    R* Ub = U + (int64_t)b * T * m;

    if (mode == kModeRollout) {
        const bool active = (g == 0);
        const R J = horizon_pass<R, NS, NC, false, KDIRECT>(D, P, homog, tid, nth, active, 0, a, 1, x0 + (int64_t)b * n,
                                                            nullptr, Ub, nullptr, nullptr, (R)0.0, Xb, nullptr, lds);
        if (active && a == 0) Jc[b] = (double)J;
        return;
    }

    if (mode == kModeLineSearch && S.singular && S.singular[b]) {  // np.linalg.solve would have raised LinAlgError
        if (tid == 0) retire_without_gains(S, b);
        return;
    }
    const int64_t gslot = (mode == kModeLineSearch && !S.gains_by_item) ? slot : b;
    const R* Kb = K + gslot * T * m * n;
    const R* db = d + gslot * T * m;
    const bool active = (g < ngrp);
    const R alpha = active ? (R)alphas[g] : (R)0.0;
    // candidate trajectories: slot of this item in the scratch (solve) or the caller's buffers (API)
    const int64_t cslot = (mode == kModeLineSearch) ? slot : b;
    R* Xw = active ? Xc + (cslot * ngrp + g) * (int64_t)(T + 1) * n : nullptr;
    R* Uw = active ? Uc + (cslot * ngrp + g) * (int64_t)T * m : nullptr;
    const R J = horizon_pass<R, NS, NC, true, KDIRECT, PIPE>(D, P, homog, tid, nth, active, g, a, ngrp, Xb, Xb, Ub, Kb, db, alpha,
                                                             Xw, Uw, lds);
    if (mode == kModeCandidates) {
        if (active && a == 0) Jc[(int64_t)b * ngrp + g] = (double)J;
        return;
    }

    // ---- one iLQR iteration's line-search decision + bookkeeping (control.py:179-211)
    int* ctl = reinterpret_cast<int*>(lds + O.ctl);
    if (active && a == 0) lds[O.J + g] = J;
    // hand-off that also makes every candidate's trajectory stores visible to the threads that copy them
    if (single_wave) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); else __syncthreads();
    if (tid == 0) {
        const int iter = S.n_bwd[b];   // this item's own iteration index (items join the batch at different times)
        const double J_star = S.J_star[b];
        int acc = -1;
        for (int i = 0; i < ngrp; ++i)
            if ((double)lds[O.J + i] < J_star) { acc = i; break; }  // strict <, NaN rejects (control.py:183)
        const int n_eval = (acc >= 0) ? acc + 1 : ngrp;
        const double J_last = (double)lds[O.J + n_eval - 1];  // last EVALUATED cost (quirk Q2)
        const double mu_before = S.mu[b];
        int status = DPILQR_STATUS_ACTIVE;
        double J_new = J_star;
        if (acc >= 0) {
            const bool conv = fabs((J_star - J_last) / J_star) < S.tol;  // control.py:184
            J_new = J_last;
            // _decrease_regularization, control.py:232-237
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
        ctl[0] = acc;
    }
    if (single_wave) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); else __syncthreads();
    const int acc = ctl[0];
    if (acc < 0) return;
    // accepted: X, U <- the accepted candidate's trajectory (a coalesced copy out of the scratch)
    const R* Xa = Xc + (cslot * ngrp + acc) * (int64_t)(T + 1) * n;
    const R* Ua = Uc + (cslot * ngrp + acc) * (int64_t)T * m;
    for (int e = tid; e < (T + 1) * n; e += nth) Xb[e] = Xa[e];
    for (int e = tid; e < T * m; e += nth) Ub[e] = Ua[e];
}

// ---- small batched entry points ------------------------------------------------------------------

// GameCost.__call__ (cost.py:197-206) for one (item, point) per thread
template <int NS, int NC>
__global__ void k_cost_eval(dpilqr_batch_desc D, int n_pts, const double* __restrict__ x, const double* __restrict__ u,
                            int terminal, double* __restrict__ cost) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)D.B * n_pts) return;
    const int b = (int)(idx / n_pts);
    const int k = D.k, n = k * NS, m = k * NC;
    const ItemParams P = item_params(D, b);
    const bool homog = homogeneous_ndims(P.n_dims, k);
    const double* xp = x + idx * n;
    const double* up = u + idx * m;
    double ref = 0.0, prox = 0.0;
    for (int a = 0; a < k; ++a) {
        double xa[NS], ua[NC];
#pragma unroll
        for (int i = 0; i < NS; ++i) xa[i] = xp[a * NS + i];
#pragma unroll
        for (int i = 0; i < NC; ++i) ua[i] = terminal ? 0.0 : up[a * NC + i];
        ref += ref_cost<NS, NC>(xa, ua, P.xf + a * NS, (terminal ? P.Qf : P.Q) + a * NS * NS, P.R + a * NC * NC,
                                terminal != 0);
    }
    for (int i = 0; i < k; ++i)
        for (int j = i + 1; j < k; ++j) {
            const int nd = homog ? 2 : min(P.n_dims[i], P.n_dims[j]);
            prox += pair_cost(xp + i * NS, xp + j * NS, nd, P.radius);
        }
    cost[idx] = D.w_prox * prox + D.w_ref * ref;
}

template <int NS, int NC, int OP>  // OP 0: f, 1: integrate, 2: linearize
__global__ void k_model_op(int n_agents, const int32_t* __restrict__ model, const double* __restrict__ x,
                           const double* __restrict__ u, double dt, double* __restrict__ o1, double* __restrict__ o2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_agents) return;
    double xa[NS], ua[NC];
#pragma unroll
    for (int j = 0; j < NS; ++j) xa[j] = x[(int64_t)i * NS + j];
#pragma unroll
    for (int j = 0; j < NC; ++j) ua[j] = u[(int64_t)i * NC + j];
    if (OP == 2) {
        double A[NS * NS], Bm[NS * NC];
        linearize_rt<NS>(model[i], xa, ua, dt, A, Bm);
#pragma unroll
        for (int j = 0; j < NS * NS; ++j) o1[(int64_t)i * NS * NS + j] = A[j];
#pragma unroll
        for (int j = 0; j < NS * NC; ++j) o2[(int64_t)i * NS * NC + j] = Bm[j];
    } else {
        double r[NS];
        if (OP == 0) f_rt<NS>(model[i], xa, ua, r);
        else integrate_rt<NS>(model[i], xa, ua, dt, r);
#pragma unroll
        for (int j = 0; j < NS; ++j) o1[(int64_t)i * NS + j] = r[j];
    }
}

// define_inter_graph_threshold (distributed.py:224-247): thread per (scenario, pair)
static __global__ void k_pairwise_graph(int S, int N, int k, int n_s, const double* __restrict__ X,
                                 const double* __restrict__ radius, int32_t* __restrict__ adj) {
    const int npairs = k * (k - 1) / 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)S * (npairs + k);
    if (idx >= total) return;
    const int s = (int)(idx / (npairs + k));
    const int p = (int)(idx - (int64_t)s * (npairs + k));
    int32_t* A = adj + (int64_t)s * k * k;
    if (p >= npairs) {  // self loops: graph[id] always contains id (distributed.py:238)
        const int i = p - npairs;
        A[i * k + i] = 1;
        return;
    }
    int i = 0, rem = p;
    while (rem >= k - 1 - i) { rem -= k - 1 - i; ++i; }
    const int j = i + 1 + rem;
    const double thr = 2 * radius[s];              // planning_radii = 2 * radius (:229)
    const int step = (N / 10 > 1) ? N / 10 : 1;    // sample_step = max(N // n_samples, 1) (:233-235)
    const double* Xs = X + (int64_t)s * N * k * n_s;
    int hit = 0;
    for (int r = 0; r < N; r += step) {            // slice(0, N+1, step) clipped to N rows
        const double* row = Xs + (int64_t)r * k * n_s;
        const double dx = row[i * n_s] - row[j * n_s], dy = row[i * n_s + 1] - row[j * n_s + 1];
        if (sqrt(dx * dx + dy * dy) < thr) { hit = 1; break; }
    }
    A[i * k + j] = hit;
    A[j * k + i] = hit;
}

}  // namespace dpilqr
