// riccati_big.hpp -- K2 for large clusters (n_x > 60, BASELINE config 5: 20 twelve-state agents, n_x = 240, n_u = 80)
// and for the fp32 arm of config 5's tolerance study: the Riccati backward sweep (ilqrSolver._backward_pass,
// control.py:116-148) with one workgroup per sub-problem, FUSED with the tile producer.
//
// At n_x = 240 nothing of the wavefront / workgroup sweeps' design survives: P alone is 460 KB (LDS: 160 KB), a dense
// tile record would be 1.28 MB per time step (193 MB per sub-problem pass).  So
//   * the per-step linearisation and quadraticisation (GameCost.quadraticize cost.py:208-239, MultiDynamicalModel.
//     linearize dynamics.py:173-186) are evaluated inside the sweep, from (X[t], U[t]) and the batch descriptor: no tile
//     records exist on this path (SURVEY 8(d) "fused variant": the sweep reads 8 (T+1) n_x + 8 T n_u bytes of
//     trajectory per pass).  Recognised plugin types only -- the same condition as for dpilqr_solve_batch;
//   * P, V = Q_xx .. P', [Q_ux | Q_u], [K | d], T3^T and Q_uu live in a per-workgroup scratch in global memory
//     (1.4 MB at n_x = 240: L2 / Infinity-Cache resident while the workgroup sweeps) and LDS holds the per-agent
//     [A_i | B_i] blocks, the pair derivatives, the symmetrised weights and the LU of Q_uu;
//   * A, B are block diagonal (uniform_block_diag, util.py:229-236), so A^T P A, B^T (P + mu I) A, B^T (P + mu I) B are
//     k^2 small block products (S1); the skipped terms are exact zeros, the association is the reference's
//     ((A^T P) A);
//   * Q_uu is factorised in LDS with partial pivoting (dgetf2 order, np.linalg.solve control.py:141-142), the
//     n_x + 1 right-hand sides are then solved one per thread by blocked substitution (S3);
//   * the dense products K^T Q_uu K, K^T Q_ux (S4, S5) are 16x16x4 MFMA tiles with operands read straight from the
//     scratch (rows = reduction index, the MFMA operand order), fp64 or fp32.
// Arithmetic type R: double (the product) or float (the tolerance study); descriptor data are fp64 and rounded on use.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "cost.hpp"
#include "models.hpp"

namespace dpilqr {

#ifndef DPILQR_BIG_THREADS
#define DPILQR_BIG_THREADS 1024
#endif
constexpr int kBigThreads = DPILQR_BIG_THREADS;   // sixteen wavefronts per sub-problem: the phases below are latency-bound loops over L2-resident data

__host__ __device__ constexpr int big_round_up(int x, int q) { return (x + q - 1) / q * q; }

// 16x16x4 matrix-pipe tile: lane (g = lane / 16, c = lane % 16) supplies A = X[l0 + g][i0 + c], B = Y[l0 + g][j0 + c]
// and owns four elements of D = X^T Y in column j0 + c; their rows differ between the two instructions.
// Value of another lane of the same 16-lane row, by DPP (one vector-pipe operation per 32 bits instead of a trip through the
// LDS crossbar): CTRL 0xB1 / 0x4E = quad_perm [1,0,3,2] / [2,3,0,1], 0x141 = row_half_mirror, 0x140 = row_mirror.  Applied in
// that order to a symmetric reduction they make every lane of the row hold the result.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_val(double v) {
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
template <int CTRL>
__device__ __forceinline__ float dpp_val(float v) { return __int_as_float(dpp_i32<CTRL>(__float_as_int(v))); }

template <typename R> struct Mfma;
template <> struct Mfma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    __device__ static __forceinline__ acc_t mac(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int v, int g) { return g + 4 * v; }    // scripts/ubench/mfma_f64.hip
    __device__ static constexpr int reg_of(int row) { return row / 4; }              // ... and its inverse: which register,
    __device__ static constexpr int group_of(int row) { return row % 4; }            // which 16-lane group holds a tile row
};
template <> struct Mfma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    __device__ static __forceinline__ acc_t mac(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int v, int g) { return 4 * g + v; }
    __device__ static constexpr int reg_of(int row) { return row % 4; }
    __device__ static constexpr int group_of(int row) { return row / 4; }
};

// An fp32 16x16x4 accumulator (lane (g, c): rows 4 g + v of column c in register v) into the fp64 instruction's arrangement (rows
// g + 4 v), which is also the matrix pipe's OPERAND arrangement (lane group g supplies reduction row 4 q + g at step q): a 4 x 4
// transposition between the registers and the 16-lane rows, four of gfx950's row swaps (v_permlane32_swap exchanges rows 2, 3 of
// its first operand with rows 0, 1 of its second, v_permlane16_swap rows 1, 3 with rows 0, 2: scripts/ubench/permlane_swap.hip).
// With it a product's result is the next product's operand without leaving the registers in fp32 too (riccati_big_s1.inc).
__device__ __forceinline__ void mfma_rows_to_operand_order(float __attribute__((ext_vector_type(4)))& t) {
    unsigned r0 = __float_as_uint(t[0]), r1 = __float_as_uint(t[1]), r2 = __float_as_uint(t[2]), r3 = __float_as_uint(t[3]);
    const auto a = __builtin_amdgcn_permlane32_swap(r0, r2, false, false); r0 = a[0]; r2 = a[1];
    const auto b = __builtin_amdgcn_permlane32_swap(r1, r3, false, false); r1 = b[0]; r3 = b[1];
    const auto c = __builtin_amdgcn_permlane16_swap(r0, r1, false, false); r0 = c[0]; r1 = c[1];
    const auto d = __builtin_amdgcn_permlane16_swap(r2, r3, false, false); r2 = d[0]; r3 = d[1];
    t[0] = __uint_as_float(r0); t[1] = __uint_as_float(r1); t[2] = __uint_as_float(r2); t[3] = __uint_as_float(r3);
}
__device__ __forceinline__ void mfma_rows_to_operand_order(double __attribute__((ext_vector_type(4)))&) {}      // (fp64: it is that order)

// the value lane `src` holds (ds_bpermute: any lane to any lane).  (Measured and dropped, round 6: for the substitution's broadcasts --
// the same lane of another 16-lane row -- gfx950's row swaps, v_permlane32_swap + v_permlane16_swap on two copies of the value
// (semantics: scripts/ubench/permlane_swap.hip), instead of the trip through the LDS crossbar: bit-identical and SLOWER, the
// substitution 61 k -> 68 k clocks per step, 256 items 55.7 -> 57.5 ms.)
__device__ __forceinline__ double lane_bcast(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v)));
}
__device__ __forceinline__ float lane_bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v))); }

// the value lane `src` holds, `src` the same in every lane (v_readlane: no trip through the LDS crossbar)
__device__ __forceinline__ double lane_get(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ float lane_get(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

// wave-wide maximum of an unsigned key, the same in every lane: DPP inside the 16-lane rows, row_bcast across them (twelve vector
// instructions and a v_readlane; no trip through the LDS crossbar)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define DPILQR_UMAX_DPP(CTRL, ROWS, BC) v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, BC));
    DPILQR_UMAX_DPP(0xB1, 0xf, true) DPILQR_UMAX_DPP(0x4E, 0xf, true) DPILQR_UMAX_DPP(0x141, 0xf, true) DPILQR_UMAX_DPP(0x140, 0xf, true)
    DPILQR_UMAX_DPP(0x142, 0xa, false)       // row_bcast:15 -> rows 1 and 3
    DPILQR_UMAX_DPP(0x143, 0xc, false)       // row_bcast:31 -> rows 2 and 3: lane 63 has seen every lane
#undef DPILQR_UMAX_DPP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// magnitude of a finite or infinite value as unsigned keys that order like it (a NaN: 0, 0 -- it never wins a maximum)
__device__ __forceinline__ void mag_keys(double x, unsigned& hi, unsigned& lo) {
    const bool num = x == x;
    hi = num ? ((unsigned)__double2hiint(x) & 0x7fffffffu) : 0u;
    lo = num ? (unsigned)__double2loint(x) : 0u;
}
__device__ __forceinline__ void mag_keys(float x, unsigned& hi, unsigned& lo) {
    hi = (x == x) ? ((unsigned)__float_as_int(x) & 0x7fffffffu) : 0u;
    lo = 0u;
}

// Scratch of one workgroup in global memory, in elements of R.  Every matrix has the same leading dimension ldw (a
// multiple of 16 >= n + 1, so that a 16-wide operand tile never leaves its row) and the control-indexed ones have mk
// rows (m rounded up to the 16-row blocks of the substitution); padding is zeroed once by the host and never written.
struct BigScratch {
    int n, m, n1, ldw, mk;
    int64_t oP, oV, oG, oKd, oT3, oQuu, oSync, oStage, stage_elems, oLU, lu_elems, total;
    __host__ __device__ BigScratch(int n_, int m_) : n(n_), m(m_) {
        n1 = n + 1;
        ldw = big_round_up(n1, 16);
        mk = big_round_up(m, 16);
        int64_t o = 0;
        oP = o;   o += (int64_t)n * ldw;
        oV = o;   o += (int64_t)n * ldw;
        oG = o;   o += (int64_t)mk * ldw;
        oKd = o;  o += (int64_t)mk * ldw;
        oT3 = o;  o += (int64_t)mk * ldw;
        oQuu = o; o += (int64_t)mk * mk;
        oSync = (o + 3) & ~(int64_t)3; o = oSync + 32;   // the team's words (k_riccati_big, nparts > 1), 16-byte aligned in either type
        // twelve-state clusters (3 m == n: the only family with four controls per twelve states): two buffers for a step's plugin
        // data -- [A|B] blocks, x - x_f, u, pair derivatives and their per-agent sums -- which a helper workgroup of the team
        // evaluates a step ahead for the main one (k_riccati_big, `staged`)
        stage_elems = 0;
        if (3 * m == n && n % 12 == 0) {
            const int k12 = n / 12, np = k12 * (k12 - 1) / 2;
            stage_elems = ((int64_t)(16 * n + n + m + 12 * np + 12 * k12 + 16) + 15) & ~(int64_t)15;
        }
        oStage = o; o += 2 * stage_elems;
        // ... and, for the same clusters, the LU factors (m_k x (m_k + 2)) and the row permutation, which the main workgroup hands to
        // the team for the substitution's column tiles
        lu_elems = stage_elems > 0 ? (((int64_t)mk * (mk + 2) + mk + 16 + 15) & ~(int64_t)15) : 0;
        oLU = o; o += lu_elems;
        total = (o + 31) & ~(int64_t)31;
    }
};

struct BigLds {   // offsets in elements of R (all even)
    int AB, QQ, RR, E, U, G3, H, HS, GS, p, LU, inv, perm, ldlu, total;
    __host__ __device__ BigLds(int k, int ns, int nc) {
        const int n = k * ns, m = k * nc, np = k * (k - 1) / 2, mk = big_round_up(m, 16);
        auto ev = [](int x) { return (x + 1) & ~1; };
        int o = 0;
        AB = o;  o += ev(k * ns * (ns + nc));   // per agent, row l: [A_i[l][:] | B_i[l][:]]
        QQ = o;  o += ev(k * ns * ns);          // w_ref (Q + Q^T) per agent (Q_f for the terminal step)
        RR = o;  o += ev(k * nc * nc);          // w_ref (R + R^T)
        E = o;   o += ev(n);                    // x - x_f
        U = o;   o += ev(m);
        G3 = o;  o += ev(np * 3);               // pair gradients
        H = o;   o += ev(np * 9);               // pair Hessians
        HS = o;  o += ev(k * 9);                // ... summed per agent over its pairs (the diagonal blocks of L_xx), in the loop's order
        GS = o;  o += ev(k * 3);                // pair gradients summed per agent, signs as in cost.py:160-163
        p = o;   o += ev(n);
        ldlu = mk + 2;
        {   // also the Jacobians' scratch between steps and the tile-transposition buffer of S5 (16 wavefronts x 2 x 16 x 17)
            const int a = mk * ldlu, b = k * ns * (ns + nc), c = 16 * 544;
            LU = o;  o += ev(a > b ? (a > c ? a : c) : (b > c ? b : c));
        }
        inv = o; o += mk;
        perm = o; o += 2 * mk + 4;              // int32 perm[mk], piv[mk], flags, in R-sized slots (>= 4 bytes each)
        total = ev(o);
    }
};

// ---- a TEAM of workgroups per item (few items: BASELINE config 5 is ONE problem) -------------------------------------------
// With fewer items than CUs the item's workgroup has the chip to itself and S5 + S6 -- 42 % of a step, bound by ONE CU's matrix
// pipes -- is the first thing more CUs can take: `nparts - 1` helper workgroups per item run their share of the tile pairs, whose
// operands and results all live in the item's global scratch.  Per step two hand-overs through words in that scratch: the main
// workgroup publishes "K, T3^T, Q ready" (a release store of the step's sequence number), every part adds itself to a counter
// when its tiles of P are stored (release), the main workgroup goes on when the counter says all have (acquire).  Agent scope:
// the parts may sit on different XCDs, whose L2s are not coherent without it.
// No hang by construction: helpers REPORT (a counter) when they start; the main workgroup decides at its first S5 whether all of
// them have (mode 1: team) or not (mode 2: alone, exactly the single-workgroup pass -- helpers that arrive later leave at once), so
// a chip busy with other work costs the speed-up, not the result; and every wait is bounded.
// No abort either: a part whose wait expires GIVES UP -- it raises every counter of the team past any target (so every other
// part's wait ends, as failed) and leaves; the main workgroup then marks the item (singular[b] = 2, which the line search turns
// into DPILQR_STATUS_FAULT and dpilqr_solve_batch into DPILQR_EHIP; the gain offsets d of the item are NaN for callers of the bare
// pass) and leaves too.  The launch ends normally and the HIP context stays usable (include/dpilqr_hip.h: "never aborts").
struct BigTeam { int flag_k, done, joined, mode, done1, done4, staged, flag_lu, doneK, xccs; };     // ten words per item, zeroed by the launcher
constexpr int kBigSpinLog2 = 22;       // polls per wait before giving up (tests lower it: tu_big.hip, DPILQR_BIG_SPIN_LOG2)
constexpr int kBigGaveUp = 1 << 30;    // a counter at or above this: some part of the team gave up

__device__ __forceinline__ int big_ld(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// thread 0 waits until *p >= target (bounded), then ITS wavefront invalidates the CU's vector cache -- one agent-scope acquire per
// workgroup, completed (s_waitcnt) before the barrier lets the other wavefronts load (the recipe of the hand-off guide; round 5
// had every one of the sixteen wavefronts fence behind the barrier: sixteen invalidates of the same cache, 2 - 4 x the time).
// false: the wait expired, or another part gave up
__device__ __forceinline__ bool big_wait_ge(int* p, int target, int* s_ok, int spin_log2) {
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) {
            int it = 0, ok = 1, v;
            while ((v = big_ld(p)) < target) {      // (relaxed: an acquire here invalidates the caches at every poll -- measured, 2 .. 5 x slower passes)
                __builtin_amdgcn_s_sleep(4);
                if ((++it >> spin_log2) != 0) { ok = 0; break; }
            }
            if (v >= kBigGaveUp) ok = 0;
            *s_ok = ok;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return *s_ok != 0;
}
// a part's arrival / the main workgroup's word, behind every wavefront's `s_waitcnt vmcnt(0)` and the workgroup barrier.
// Two kinds of RELEASE (round 6), chosen ONCE per pass by the main workgroup from what the parts report:
//   agent scope -- the XCD's L2 written back (buffer_wbl2 sc1): right wherever the parts sit;
//   local       -- every part of the team has reported the SAME XCD as the main workgroup (HW_REG_XCC_ID, at the join): the parts
//     share one L2, the point of coherence of their CUs' write-through vector caches, so a release is the stores' completion
//     alone (the s_waitcnt before the barrier) and the write-back of the whole L2 -- every team's dirty lines on that XCD, at
//     every one of a step's seven hand-overs -- is skipped.  One item 18.4 -> 17.7 ms, 32 items (four teams per XCD) 22.9 -> 20.1.
//     Nothing is assumed about the dispatcher: the launch's numbering puts a team on one XCD when workgroups are dealt to the
//     XCDs in turn, and when they are not, the parts' reports differ and the pass runs on agent-scope releases.
// The ACQUIRE stays `buffer_inv sc1` in both: measured (scripts/ubench/l1_inv.hip, profiles/r06_l1_inv.txt), `buffer_inv sc0` --
// workgroup scope, the memory model's acquire between the CUs of a threadgroup-split workgroup -- leaves a CU's vector cache as it
// is outside that mode (every re-read word stale), and loads marked sc0 hit the stale lines too; `buffer_inv sc1` drops the
// vector cache and leaves the L2's dirty lines in place (the re-read costs what an L2 hit costs).  A first version of the
// local hand-over with `buffer_inv sc0` passed every fp64 check and failed sporadically in fp32 at 32 items -- stale lines
// survive where less data streams through the 32 KB cache.
__device__ __forceinline__ void big_arrive(int* p, bool local) {
    if (local) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(p, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void big_publish(int* p, int v, bool local) {
    if (local) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int big_xcc() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}

// a part gives up: every wait of the team, current or future, ends as failed; helpers that have not looked yet see mode 3 and leave
__device__ __forceinline__ void big_give_up(BigTeam* team) {
    if (threadIdx.x == 0) {
        __hip_atomic_store(&team->flag_k, kBigGaveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&team->done, kBigGaveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&team->done1, kBigGaveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&team->done4, kBigGaveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&team->staged, kBigGaveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&team->flag_lu, kBigGaveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&team->doneK, kBigGaveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&team->mode, 3, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename R>
__global__ void k_big_team_reset(R* scratch_all, int64_t stride, int64_t o_sync, int n_slots) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    BigTeam* team = reinterpret_cast<BigTeam*>(scratch_all + (int64_t)s * stride + o_sync);
    team->flag_k = 0; team->done = 0; team->joined = 0; team->mode = 0; team->done1 = 0; team->done4 = 0; team->staged = 0; team->flag_lu = 0; team->doneK = 0; team->xccs = 0;
}

template <typename R, int NS, int NC>
__global__ __launch_bounds__(kBigThreads) void k_riccati_big(dpilqr_batch_desc D, const R* __restrict__ X,
                                                             const R* __restrict__ U, const double* __restrict__ mu_arr,
                                                             R* __restrict__ Kout, R* __restrict__ dout,
                                                             int32_t* __restrict__ singular,
                                                             const int32_t* __restrict__ items,
                                                             const int32_t* __restrict__ n_items, int gains_by_item,
                                                             R* scratch_all, int n_slots, int nparts, int team_dbg) {
    typedef typename Mfma<R>::acc_t acc_t;
    constexpr int NSC = NS + NC;
    // nparts > 1: workgroup x + 8 y is part y % nparts of slot x + 8 (y / nparts) -- workgroups are dealt to the eight XCDs in turn,
    // so an item's parts share an L2 (a placement the hand-overs do not rely on)
    const int wg = blockIdx.x;
    // (readfirstlane: the division runs on the vector pipe, and every pointer derived from a per-lane slot would be per lane)
    const int slot = __builtin_amdgcn_readfirstlane(nparts > 1 ? (wg & 7) + 8 * ((wg >> 3) / nparts) : wg);
    const int part = __builtin_amdgcn_readfirstlane(nparts > 1 ? (wg >> 3) % nparts : 0);
    if (slot >= n_slots) return;
    if (n_items && slot >= *n_items) return;
    // (the item index is the same in every lane: say so, or every pointer derived from it -- scratch regions, trajectory, gains,
    // descriptor arrays: three dozen 64-bit values -- is held per lane and, at 128 registers per lane, spilled)
    const int b = __builtin_amdgcn_readfirstlane(items ? items[slot] : slot);
    if (b >= D.B) return;
    const int64_t gslot = gains_by_item ? b : slot;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g16 = lane >> 4, c16 = lane & 15;
    const int k = D.k, T = D.T, n = k * NS, m = k * NC, npairs = k * (k - 1) / 2;
    const BigScratch S(n, m);
    const BigLds O(k, NS, NC);
    const int n1 = S.n1, ldw = S.ldw, mk = S.mk, ldlu = O.ldlu;
    const ItemParams P = item_params(D, b);
    const R mu = (R)mu_arr[b], wr = (R)D.w_ref, wp = (R)D.w_prox, radius = (R)P.radius, dt = (R)D.dt;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    R* lds = reinterpret_cast<R*>(lds_raw);
    R* sAB = lds + O.AB; R* sQQ = lds + O.QQ; R* sRR = lds + O.RR; R* sE = lds + O.E; R* sU = lds + O.U;
    R* sG3 = lds + O.G3; R* sH = lds + O.H; R* sHS = lds + O.HS; R* sGS = lds + O.GS; R* sp = lds + O.p; R* sLU = lds + O.LU; R* sInv = lds + O.inv;
    int* sPerm = reinterpret_cast<int*>(lds + O.perm);
    int* sPiv = sPerm + mk;
    int* sFlag = sPiv + mk;      // [0]: singular, [1]: pivot row of the current column

    R* gP = scratch_all + (int64_t)slot * S.total + S.oP;
    // Twelve-state fp64 clusters: V shares P's array.  Their S1 rewrites a block pair's 12 x 12 block in the wavefront that read it
    // (all of it, into registers, before its first store) and S5 a tile pair likewise, so V = Q_xx .. never needs a place of its
    // own -- and a third less scratch is touched per item (0.94 instead of 1.43 MB: 256 items then fit the 256 MB Infinity Cache).
    // Column n of the shared array is Q_x after S1 and the new p after S5; P itself never uses it.  The other forms of S1
    // (work items of several wavefronts read one block) keep V apart.
    constexpr bool kAliasPV = (sizeof(R) == 8 && NS == 12 && NC == 4);
    const int64_t oVa = kAliasPV ? S.oP : S.oV;
    R* gV = scratch_all + (int64_t)slot * S.total + oVa;
    R* gG = scratch_all + (int64_t)slot * S.total + S.oG;
    R* gKd = scratch_all + (int64_t)slot * S.total + S.oKd;
    R* gT3 = scratch_all + (int64_t)slot * S.total + S.oT3;
    R* gQuu = scratch_all + (int64_t)slot * S.total + S.oQuu;
    const R* Xb = X + (int64_t)b * (T + 1) * n;
    const R* Ub = U + (int64_t)b * T * m;

    // ---- once per pass: w_ref (R + R^T); the LU buffer's padding is the identity
    for (int e = tid; e < k * NC * NC; e += kBigThreads) {
        const int a = e / (NC * NC), r = e - a * NC * NC, li = r / NC, lj = r - li * NC;
        const double* Rm = P.R + a * NC * NC;
        sRR[e] = wr * ((R)Rm[li * NC + lj] + (R)Rm[lj * NC + li]);
    }
    for (int e = tid; e < mk * ldlu; e += kBigThreads) {
        const int r = e / ldlu, c = e - r * ldlu;
        sLU[e] = (r == c && r >= m) ? (R)1.0 : (R)0.0;
    }
    if (tid < 2) sFlag[tid] = 0;

    // per-step plugin evaluation at (x, u) of step t: linearisation, x - x_f, pair derivatives, weights
    auto stage_step = [&](int t, bool terminal) {
        // (its own thread id, opaque to the optimiser like the phases' lane terms below: hoisted out of the horizon loop the
        // staging loops' per-lane addresses stayed live through every phase and were spilled)
        int tid_s_ = threadIdx.x;
        asm volatile("" : "+v"(tid_s_));
        const int tid = tid_s_;
        const R* xt = Xb + (int64_t)t * n;
        const R* ut = Ub + (int64_t)(terminal ? 0 : t) * m;
        // the Jacobians are written straight into LDS (a Quadcopter12D Jacobian held in registers costs 400 of them and
        // with that every other phase its occupancy): per agent a scratch block [A | B] inside the LU buffer, which is dead
        // between the steps, interleaved into sAB below
        R* sJ = sLU;
        for (int a = tid; a < k; a += kBigThreads) {
            R x[NS], u[NC];
#pragma unroll
            for (int i = 0; i < NS; ++i) x[i] = xt[a * NS + i];
#pragma unroll
            for (int i = 0; i < NC; ++i) u[i] = terminal ? (R)0.0 : ut[a * NC + i];
            if (!terminal) linearize_rt<NS>(P.model[a], x, u, dt, sJ + a * NS * NSC, sJ + a * NS * NSC + NS * NS);
#pragma unroll
            for (int i = 0; i < NS; ++i) sE[a * NS + i] = x[i] - (R)P.xf[a * NS + i];
#pragma unroll
            for (int i = 0; i < NC; ++i) sU[a * NC + i] = u[i];
        }
        for (int p = tid; p < npairs; p += kBigThreads) {   // pairs in itertools.combinations order
            int i = 0, rem = p;
            while (rem >= k - 1 - i) { rem -= k - 1 - i; ++i; }
            const int j = i + 1 + rem;
            const int nd = min(P.n_dims[i], P.n_dims[j]);  // cost.py:145
            R gg[3], HH[9];
            pair_quadraticize(xt + i * NS, xt + j * NS, nd, radius, gg, HH);
#pragma unroll
            for (int c = 0; c < 3; ++c) sG3[p * 3 + c] = gg[c];
#pragma unroll
            for (int c = 0; c < 9; ++c) sH[p * 9 + c] = HH[c];
        }
        __syncthreads();
        // per agent: the sum of its pairs' Hessians / signed gradients, in the order the per-element loops took them (o ascending
        // from 0.0): what the diagonal blocks of L_xx and l_x add -- once per step instead of once per element
        for (int e = tid; e < k * 12; e += kBigThreads) {
            const int a = e / 12, c = e - a * 12;
            R acc = 0.0;
            if (c < 9) {
                for (int o = 0; o < k; ++o) {
                    if (o == a) continue;
                    acc += sH[((o < a) ? pair_index(o, a, k) : pair_index(a, o, k)) * 9 + c];
                }
                sHS[a * 9 + c] = acc;
            } else {
                const int lj = c - 9;
                for (int o = 0; o < k; ++o) {
                    if (o == a) continue;
                    if (o < a) acc += -sG3[pair_index(o, a, k) * 3 + lj];
                    else       acc += sG3[pair_index(a, o, k) * 3 + lj];
                }
                sGS[a * 3 + lj] = acc;
            }
        }
        if (!terminal) {
            for (int e = tid; e < k * NS * NSC; e += kBigThreads) {
                const int a = e / (NS * NSC), q = e - a * NS * NSC, l = q / NSC, i = q - l * NSC;
                sAB[e] = (i < NS) ? sJ[a * NS * NSC + l * NS + i] : sJ[a * NS * NSC + NS * NS + l * NC + (i - NS)];
            }
            __syncthreads();
            // the LU buffer again: zero inside, identity on the padding's diagonal
            for (int e = tid; e < mk * ldlu; e += kBigThreads) {
                const int r = e / ldlu, c = e - r * ldlu;
                sLU[e] = (r == c && r >= m) ? (R)1.0 : (R)0.0;
            }
        }
        if (t == T || t == T - 1) {   // the weights change once: Q_f for the terminal record, Q for all others
            for (int e = tid; e < k * NS * NS; e += kBigThreads) {
                const int a = e / (NS * NS), r = e - a * NS * NS, li = r / NS, lj = r - li * NS;
                const double* M = (terminal ? P.Qf : P.Q) + a * NS * NS;
                sQQ[e] = wr * ((R)M[li * NS + lj] + (R)M[lj * NS + li]);
            }
        }
    };
    // L_xx[gi][gj], L_x[gi] of the staged step (cost.py:228-237, 160-169; same association as tiles.hpp)
    auto lxx = [&](int ai, int li, int aj, int lj) -> R {
        R v = 0.0;
        if (ai == aj) v = sQQ[ai * NS * NS + li * NS + lj];
        if (k > 1 && li < 3 && lj < 3) {
            R acc = 0.0;
            if (ai == aj) {
                acc = sHS[ai * 9 + li * 3 + lj];
            } else {
                const int p = (ai < aj) ? pair_index(ai, aj, k) : pair_index(aj, ai, k);
                acc += -sH[p * 9 + li * 3 + lj];
            }
            v += wp * acc;
        }
        return v;
    };
    auto lx = [&](int a, int lj) -> R {
        R v = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i) v += sE[a * NS + i] * sQQ[a * NS * NS + i * NS + lj];
        if (k > 1 && lj < 3) {
            v += wp * sGS[a * 3 + lj];
        }
        return v;
    };

    // ---- terminal condition: p = l_x(T), P = l_xx(T)   (control.py:125-129)
    __syncthreads();
    if (part == 0) {     // (the main workgroup; a helper goes straight to its loop below)
    stage_step(T, true);
    __syncthreads();
    for (int e = tid; e < n * n; e += kBigThreads) {
        const int i = e / n, j = e - i * n;
        gP[(int64_t)i * ldw + j] = lxx(i / NS, i % NS, j / NS, j % NS);
    }
    for (int i = tid; i < n; i += kBigThreads) sp[i] = lx(i / NS, i % NS);
    }
    __syncthreads();

    // The lane terms of a phase (tile coordinates, operand addresses) are formed at the phase's start from a thread id the
    // optimiser cannot see through: hoisted out of the horizon loop they are some fifty 64-bit addresses per lane, and at the
    // 128 registers per lane that sixteen wavefronts per CU allow they were spilled in the prologue and reloaded inside the
    // phases' inner loops (round 2: 82 spilled registers, 324 B of scratch per lane).  What the IR-level trick cannot reach --
    // constants and addresses the MACHINE-level LICM hoists -- is switched off for this translation unit
    // (__graft_entry__.UNIT_FLAGS: -mllvm -disable-machine-licm); together with round 5's substitution: no spills.
#define BIG_LANE_TERMS()                                                                                   \
    int tid_p_ = threadIdx.x;                                                                              \
    asm volatile("" : "+v"(tid_p_));                                                                       \
    const int tid = tid_p_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g16 = lane >> 4, c16 = lane & 15; \
    (void)wave; (void)g16; (void)c16; (void)lane;
    BigTeam* team = reinterpret_cast<BigTeam*>(scratch_all + (int64_t)slot * S.total + S.oSync);
    R* const gStage = scratch_all + (int64_t)slot * S.total + S.oStage;
    const int nStage1 = O.QQ - O.AB, nStage2 = O.p - O.E;          // [A|B] blocks ; x - x_f, u, pair derivatives, their sums
    const bool staged_ok = S.stage_elems >= nStage1 + nStage2;       // (the scratch holds the two buffers: twelve-state clusters)
    R* const gLU = scratch_all + (int64_t)slot * S.total + S.oLU;   // the factors and the permutation, for the team's substitution
    int* const gLUperm = reinterpret_cast<int*>(gLU + (int64_t)mk * ldlu);
    const bool lu_ok = S.lu_elems >= (int64_t)mk * ldlu + mk;
    const int spin_log2 = (team_dbg >> 8) ? (team_dbg >> 8) : kBigSpinLog2;
    // the main workgroup's way out when the team has failed: the item is marked (for callers of the bare pass, who hand over no
    // `singular`: the first gain offset of the item is NaN), the launch goes on
    auto gave_up = [&]() {
        big_give_up(team);
        if (threadIdx.x == 0) {
            if (singular) singular[b] = 2;
            dout[gslot * (int64_t)T * m] = (R)__builtin_nan("");
        }
    };
    // (the team also shares S1 where S1 is the matrix-pipe form -- twelve-state fp64, config 5; the vector forms of the other
    // instantiations keep it on the main workgroup: with the helper's copy of them the fp32 twelve-state kernel spilled 40 registers)
#ifdef DPILQR_BIG_TEAM_S1_ALL
    constexpr bool kTeamS1 = true;
#else
    constexpr bool kTeamS1 = (NS == 12 && NC == 4);
#endif
    constexpr bool kTeamSolve = kTeamS1 || (NS == 12 && NC == 4);      // ... and the substitution's column tiles (round 6; twelve-state clusters, either type)
    int coop = 0;      // 1: this pass is run by the team
    bool local = false;   // ... on hand-overs through the XCD's own L2 (big_arrive)
    if (part > 0) {    // a helper: its share of every step's tile pairs, nothing else
        // (tests: DPILQR_BIG_TEAM_LATE=1 makes the helpers report a few milliseconds late -- after the main workgroup's decision --
        // which is what a chip busy with other work does to them: the pass must then be the single workgroup's; =2 makes them join
        // and then never work -- fault injection: the main workgroup's first wait for them must expire into a status)
        if ((team_dbg & 3) == 1) for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(127);
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_or(&team->xccs, 1 << big_xcc(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // where this part runs
            __hip_atomic_fetch_add(&team->joined, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!big_wait_ge(&team->mode, 1, &sFlag[2], spin_log2)) { big_give_up(team); return; }   // (the main workgroup never decided)
        {   // 1: the team, agent-scope hand-overs; 4: the team, all on one XCD; else the main workgroup went ahead alone (2), or the team gave up (3)
            const int mode = __builtin_amdgcn_readfirstlane(big_ld(&team->mode));
            if (mode != 1 && mode != 4) return;
            local = mode == 4;
        }
        if ((team_dbg & 3) == 2) { big_wait_ge(&team->mode, 3, &sFlag[2], spin_log2 + 2); return; }   // (until the main workgroup has given up)
        if constexpr (kTeamS1) {     // w_ref (Q + Q^T): the helper's first stage is step T - 2's, the weights' own step has passed
            for (int e = threadIdx.x; e < k * NS * NS; e += kBigThreads) {
                const int a = e / (NS * NS), r = e - a * NS * NS, li = r / NS, lj = r - li * NS;
                const double* M = P.Q + a * NS * NS;
                sQQ[e] = wr * ((R)M[li * NS + lj] + (R)M[lj * NS + li]);
            }
        }
        for (int t = T - 1; t >= 0; --t) {
            if (kTeamS1 && t < T - 1) {
                // S1 of this step with the team (the first step's ran before the team was decided): P of the previous step is
                // complete when EVERY part has added itself to `done`.  The step's plugin data are in place already (below).
                if (!big_wait_ge(&team->done, (T - 1 - t) * nparts, &sFlag[2], spin_log2)) { big_give_up(team); return; }
                {
                    const int part_ = part, nparts_ = nparts;
#include "riccati_big_s1.inc"
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) big_arrive(&team->done1, local);
            }
            if (kTeamS1 && t > 0) {
                // The NEXT step's plugin data, now: they depend on (X, U) alone, and from here to the main workgroup's word
                // "K is ready" a helper has a quarter of a millisecond to spare (the LU and the substitution run there).  Part 1
                // also stores them for the main workgroup, which then loads 53 KB at the top of its next step instead of
                // evaluating twenty linearisations and 190 pair derivatives there, on the critical path (47 k clocks of a step's
                // 420 k; round 6).  Two buffers by the step's parity: the main workgroup reads step t - 1's while step t - 2's is
                // written only after every part, the main one included, has finished step t - 1's S1.
                stage_step(t - 1, false);
                __syncthreads();
                if (part == 1 && staged_ok) {
                    R* gS = gStage + (int64_t)((t - 1) & 1) * S.stage_elems;
                    int tid_c = threadIdx.x;       // (opaque: hoisted out of the horizon loop the copy's per-lane addresses were spilled)
                    asm volatile("" : "+v"(tid_c));
                    for (int e = tid_c; e < nStage1; e += kBigThreads) gS[e] = lds[O.AB + e];
                    for (int e = tid_c; e < nStage2; e += kBigThreads) gS[nStage1 + e] = lds[O.E + e];
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (threadIdx.x == 0) big_publish(&team->staged, T - (t - 1), local);
                }
            }
            if (kTeamSolve && lu_ok && t < T - 1) {
                // the substitution's column tiles (the first step's ran before the team was decided): the main workgroup's factors
                // from the scratch into this workgroup's LU buffer, this part's tiles, then every part's arrival = [K | d] complete
                if (!big_wait_ge(&team->flag_lu, T - t, &sFlag[2], spin_log2)) { big_give_up(team); return; }
                {
                    int tid_c = threadIdx.x;
                    asm volatile("" : "+v"(tid_c));
                    for (int e = tid_c; e < mk * ldlu; e += kBigThreads) sLU[e] = gLU[e];
                    if (tid_c < mk) sPerm[tid_c] = gLUperm[tid_c];
                }
                __syncthreads();
                {
                    const int part_ = part, nparts_ = nparts;
#include "riccati_big_solve.inc"
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) big_arrive(&team->doneK, local);
                if (!big_wait_ge(&team->doneK, (T - 1 - t) * nparts, &sFlag[2], spin_log2)) { big_give_up(team); return; }
            } else {
                if (!big_wait_ge(&team->flag_k, T - t, &sFlag[2], spin_log2)) { big_give_up(team); return; }
            }
            {
                const int part_ = part, nparts_ = nparts;
#include "riccati_big_s4.inc"
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) big_arrive(&team->done4, local);
            if (!big_wait_ge(&team->done4, (T - t) * nparts, &sFlag[2], spin_log2)) { big_give_up(team); return; }
            {
                const int part_ = part, nparts_ = nparts;
#include "riccati_big_pairs.inc"
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) big_arrive(&team->done, local);
        }
        return;
    }

#ifdef DPILQR_PHASE_STAMPS
    unsigned long long bph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bph_t = __builtin_amdgcn_s_memtime();
#define BPHASE(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); bph[i] += now_ - bph_t; bph_t = now_; }
#elif defined(DPILQR_PHASE_MARKS)   // assembly listings only (scripts/isa_census.py)
#define BPHASE(i) asm volatile("; ==== end of phase " #i);
#elif defined(DPILQR_BIG_STOP)      // diagnostic builds only: leave the kernel behind phase DPILQR_BIG_STOP of the first step (localising a fault)
#define BPHASE(i) if (DPILQR_BIG_STOP == (i)) return;
#else
#define BPHASE(i)
#endif
    for (int t = T - 1; t >= 0; --t) {
        if (kTeamS1 && coop && staged_ok) {
            // the team's part 1 has evaluated this step's plugin data while this workgroup factorised and substituted (the helper
            // loop above): they are loaded, not made.  No wait of its own (later in round 6): part 1 stores the data and raises
            // `staged` BEFORE it arrives at the previous step's `done`, which this workgroup has just acquired -- whatever a part did
            // before that arrival is visible here -- so a poll, an acquire and a barrier per step (~3 k clocks) told nothing new.
            // (`staged` itself remains for part 1's own bookkeeping and the give-up path.)
            const R* gS = gStage + (int64_t)(t & 1) * S.stage_elems;
            int tid_c = threadIdx.x;
            asm volatile("" : "+v"(tid_c));
            for (int e = tid_c; e < nStage1; e += kBigThreads) lds[O.AB + e] = gS[e];
            for (int e = tid_c; e < nStage2; e += kBigThreads) lds[O.E + e] = gS[nStage1 + e];
            // the LU buffer as stage_step leaves it: zero inside, identity on the padding's diagonal
            for (int e = tid_c; e < mk * ldlu; e += kBigThreads) {
                const int r = e / ldlu, c = e - r * ldlu;
                sLU[e] = (r == c && r >= m) ? (R)1.0 : (R)0.0;
            }
        } else {
            stage_step(t, false);
        }
        __syncthreads();
        BPHASE(0)

        {
            const int part_ = 0, nparts_ = (kTeamS1 && coop) ? nparts : 1;
#include "riccati_big_s1.inc"
        }
        { BIG_LANE_TERMS()
        // Q_x = l_x + A^T p -> column n of V ; Q_u = l_u + B^T p -> column n of [Q_ux | Q_u]
        for (int i = tid; i < n + m; i += kBigThreads) {
            R s = 0.0;
            if (i < n) {
                const int a = i / NS, li = i - a * NS;
#pragma unroll
                for (int l = 0; l < NS; ++l) s = fma(sAB[(a * NS + l) * NSC + li], sp[a * NS + l], s);
                gV[(int64_t)i * ldw + n] = lx(a, li) + s;
            } else {
                const int ia = i - n, a = ia / NC, lc = ia - a * NC;
#pragma unroll
                for (int l = 0; l < NS; ++l) s = fma(sAB[(a * NS + l) * NSC + NS + lc], sp[a * NS + l], s);
                R lu = 0.0;
#pragma unroll
                for (int q = 0; q < NC; ++q) lu += sU[a * NC + q] * sRR[a * NC * NC + q * NC + lc];
                gG[(int64_t)ia * ldw + n] = lu + s;
            }
        }
        }
        if (kTeamS1 && coop) {      // the team's S1: every part's block products stored, then Q_uu -- the other parts' entries of it -- into the LU buffer
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) big_arrive(&team->done1, local);
            if (!big_wait_ge(&team->done1, (T - 1 - t) * nparts, &sFlag[2], spin_log2)) { gave_up(); return; }
            for (int e = threadIdx.x; e < m * m; e += kBigThreads) {
                const int a = e / m, c = e - a * m;
                sLU[a * ldlu + c] = gQuu[(int64_t)a * mk + c];
            }
        }
        __syncthreads();
        BPHASE(1)

        { BIG_LANE_TERMS()
        // ---- S3a: LU of Q_uu in LDS, partial pivoting in dgetf2's order (first POSITION of largest magnitude), without moving
        // rows: sPerm[pos] is the row that stands at position pos after the exchanges so far; the factors stay where they are
        // and every later access goes through sPerm
        for (int r = tid; r < mk; r += kBigThreads) sPerm[r] = r;
        __syncthreads();
#if !defined(DPILQR_BIG_LU_PLAIN) && !defined(DPILQR_BIG_LU_COLUMN)
        // BLOCKED, trailing update on the matrix pipe (round 6).  Panels of FOUR columns -- the reduction depth of one 16x16x4
        // product.  Per panel: (A) the first wavefront factorises it in registers, lane = position (two per lane beyond 64
        // rows): the arg-max is a wave-wide maximum (DPP within the 16-lane rows, two exchanges across them) and a ballot -- the
        // first position that holds it, dgetf2's idamax --, the pivot row goes out by v_readlane, the exchange is two lanes
        // trading contents, the reciprocal is computed once on a value every lane holds; (B) the same wavefront forms the
        // panel's rows of U to the right, one column per lane, three dependent multiply-adds; (C) all sixteen wavefronts apply
        // the panel to the trailing block, one 16x16x4 product per tile: A = -L (a lane's position, reduction row g), B = U
        // (reduction row g, a lane's column), C = the tile as it stands.  Two workgroup barriers per panel, 40 per
        // factorisation instead of 80, and no chain of LDS round trips per column.  Same pivots (partial pivoting, first
        // position of the largest magnitude), the same reciprocal 1 / pivot, the same multipliers a * (1 / pivot), every entry
        // updated by the same multiply-adds in the same order (the matrix pipe accumulates its four reduction rows in
        // ascending order).  The multipliers are stored as such (the column form stored a and scaled at the end).
        // (The column form with its look-ahead wavefront: -DDPILQR_BIG_LU_COLUMN, A/B builds.  Round 5's blocked attempt -- sixteen-
        // column panels, the trailing block on the vector pipe one thread per row -- was slower than the column form.)
        // Panel WIDTH (later in round 6): eight columns -- two chained products per trailing tile -- wherever the panel's rows fit
        // one row per lane (m - K0 <= 64: every panel but the first sixteen columns' at n_u = 80; with two rows per lane the
        // eight-column panel does not fit the 128 registers a 1024-thread workgroup has): twelve trailing updates and 24 barriers
        // per factorisation instead of twenty and 40, and what a trailing update costs is its tiles' trips through LDS, not the
        // product.  A column's eliminations reach every entry in the same order either way -- inside the panel's registers, or
        // as reduction rows 0 .. 3 then 4 .. 7 of the chained products: bit-identical factors (-DDPILQR_BIG_LU_PANEL4: all
        // panels four wide, A/B builds).
        // (one loop per width, the width a compile-time constant in each: with the width a run-time value of ONE loop the kernel
        // spilled 107 registers)
        auto lu_step = [&](const int K0, auto pw_c) __attribute__((always_inline)) {
            constexpr int pw = decltype(pw_c)::value;
            const int nbp = min(pw, m - K0), base = K0 + nbp;
#ifdef DPILQR_PHASE_STAMPS
            const unsigned long long lu_t0 = __builtin_amdgcn_s_memtime();
#endif
            if (wave == 0) {
                // Lanes are ROWS for the whole panel and nothing moves: a row's POSITION is a label that changes hands at an
                // exchange (the row at position K0 + j takes the pivot's old position), a pivot row simply stops taking part.
                // The arg-max works on the magnitudes' bit patterns (unsigned keys order like the magnitudes): the high words'
                // wave maximum, then the low words' among the lanes that hold it -- integer DPP reductions --, then a ballot.
                // Instantiations: rows K0 .. K0 + 63 only (PW 4 or 8), or a second row per lane (PW 4).
                auto panel = [&](auto two_tag, auto pw_tag) __attribute__((always_inline)) {
                    constexpr bool TWO = decltype(two_tag)::value;
                    constexpr int PW = decltype(pw_tag)::value;
                    const int p0 = K0 + lane, p1 = p0 + 64;
                    const bool in0 = p0 < m, in1 = TWO && p1 < m;
                    const int r0 = sPerm[min(p0, mk - 1)], r1 = TWO ? sPerm[min(p1, mk - 1)] : 0;
                    int pos0 = p0, pos1 = p1;
                    bool act0 = in0, act1 = in1;
                    R a0[PW], a1[TWO ? PW : 1];
#pragma unroll
                    for (int j = 0; j < PW; ++j) {
                        a0[j] = (in0 && j < nbp) ? sLU[r0 * ldlu + K0 + j] : (R)0.0;
                        if (TWO) a1[j] = (in1 && j < nbp) ? sLU[r1 * ldlu + K0 + j] : (R)0.0;
                    }
                    int pvl[PW];                            // the pivots: lane | chunk << 6 (the same in every lane)
#pragma unroll
                    for (int j = 0; j < PW; ++j) pvl[j] = 0;
#pragma unroll
                    for (int j = 0; j < PW; ++j) {
                        if (j >= nbp) break;
                        const R x0 = a0[j], x1 = TWO ? a1[j] : (R)0.0;
                        // every lane divides for ITS candidate while the maximum is being found: the pivot's reciprocal is then
                        // one lane read away (the division, a dozen dependent instructions, used to follow the search)
                        constexpr bool EARLY = true;
                        R i0 = (R)0.0, i1 = (R)0.0;
                        if constexpr (EARLY) {
                            i0 = (x0 == (R)0.0) ? (R)0.0 : (R)1.0 / x0;
                            i1 = TWO ? ((x1 == (R)0.0) ? (R)0.0 : (R)1.0 / x1) : (R)0.0;
                            // (pinned HERE: the compiler otherwise sinks the divisions behind the search, into the branch that picks the
                            // pivot's chunk -- back onto the column's critical path)
                            asm volatile("" : "+v"(i0));
                            if (TWO) asm volatile("" : "+v"(i1));
                        }
                        unsigned h0, l0k, h1 = 0u, l1k = 0u;
                        mag_keys(x0, h0, l0k);
                        if (TWO) mag_keys(x1, h1, l1k);
                        h0 = act0 ? h0 : 0u; l0k = act0 ? l0k : 0u;
                        h1 = act1 ? h1 : 0u; l1k = act1 ? l1k : 0u;
                        const unsigned hm = wave_max_u32(TWO ? max(h0, h1) : h0);
                        bool c0 = act0 && h0 == hm, c1 = act1 && h1 == hm;
                        unsigned lm = 0u;
                        if constexpr (sizeof(R) == 8) {
                            // the low words decide only among rows that share the largest high word (sign, exponent, twenty mantissa
                            // bits): almost always ONE row holds it and the second reduction is skipped
                            const unsigned long long q0 = __builtin_amdgcn_ballot_w64(c0), q1 = TWO ? __builtin_amdgcn_ballot_w64(c1) : 0ull;
                            if (__builtin_popcountll(q0) + __builtin_popcountll(q1) > 1) {
                                lm = wave_max_u32(TWO ? max(c0 ? l0k : 0u, c1 ? l1k : 0u) : (c0 ? l0k : 0u));
                                c0 = c0 && l0k == lm; c1 = c1 && l1k == lm;
                            } else {
                                lm = (q0 | q1) != 0ull ? 1u : 0u;     // (only its being non-zero is used below, with hm)
                            }
                        }
                        const bool ok = (hm | lm) != 0u;              // a positive (or infinite) magnitude exists
                        unsigned long long k0, k1 = 0ull;
                        if (ok) {
                            k0 = __builtin_amdgcn_ballot_w64(c0);
                            if (TWO) k1 = __builtin_amdgcn_ballot_w64(c1);
                        } else {     // no positive finite candidate -- singular, or poisoned: flagged; position K0 + j keeps its row
                            k0 = __builtin_amdgcn_ballot_w64(act0 && pos0 == K0 + j);
                            if (TWO) k1 = __builtin_amdgcn_ballot_w64(act1 && pos1 == K0 + j);
                        }
                        int pc = (!TWO || k0 != 0ull) ? 0 : 1;
                        int pl = (!TWO || k0 != 0ull) ? __builtin_ctzll(k0) : __builtin_ctzll(k1);
                        if (__builtin_popcountll(k0) + __builtin_popcountll(k1) > 1) {
                            // several rows hold the maximum: dgetf2 takes the first POSITION (rare; a scalar loop over the tied lanes)
                            int bestp = 0x7fffffff;
                            for (unsigned long long w = k0; w != 0ull; w &= w - 1) {
                                const int l = __builtin_ctzll(w), pp_ = __builtin_amdgcn_readlane(pos0, l);
                                if (pp_ < bestp) { bestp = pp_; pl = l; pc = 0; }
                            }
                            for (unsigned long long w = k1; w != 0ull; w &= w - 1) {
                                const int l = __builtin_ctzll(w), pp_ = __builtin_amdgcn_readlane(pos1, l);
                                if (pp_ < bestp) { bestp = pp_; pl = l; pc = 1; }
                            }
                        }
                        pvl[j] = pl | (pc << 6);
                        // the exchange, on the labels: whoever stands at K0 + j takes the pivot's position, the pivot takes K0 + j
                        const int ppos = (TWO && pc) ? __builtin_amdgcn_readlane(pos1, pl) : __builtin_amdgcn_readlane(pos0, pl);
                        pos0 = (act0 && pos0 == K0 + j) ? ppos : pos0;
                        if (TWO) pos1 = (act1 && pos1 == K0 + j) ? ppos : pos1;
                        const bool me = lane == pl;
                        if (TWO && pc) { pos1 = me ? K0 + j : pos1; act1 = act1 && !me; }
                        else { pos0 = me ? K0 + j : pos0; act0 = act0 && !me; }
                        R inv;      // (pv == 0) ? 0 : 1 / pv
                        if constexpr (EARLY) {
                            inv = (TWO && pc) ? lane_get(i1, pl) : lane_get(i0, pl);      // the pivot lane's own
                        } else {
                            const R pv = lane_get(x0, pl);      // (a value every lane holds: the same division, the same bits)
                            inv = (pv == (R)0.0) ? (R)0.0 : (R)1.0 / pv;
                        }
                        if (lane == 0) {
                            if (!ok) sFlag[0] = 1;            // zero (or NaN) pivot: np.linalg.solve would raise
                            sInv[K0 + j] = inv;
                        }
                        // the multipliers of the rows still in play, and their entries of the panel's later columns (a row that is
                        // out of play multiplies by zero: its entries stay as they are)
                        a0[j] = act0 ? x0 * inv : x0;
                        const R m0 = act0 ? a0[j] : (R)0.0;
                        R m1 = (R)0.0;
                        if (TWO) { a1[j] = act1 ? x1 * inv : x1; m1 = act1 ? a1[j] : (R)0.0; }
#pragma unroll
                        for (int jj = j + 1; jj < PW; ++jj) {
                            const R prj = (TWO && pc) ? lane_get(a1[jj], pl) : lane_get(a0[jj], pl);
                            a0[jj] = fma(-m0, prj, a0[jj]);
                            if (TWO) a1[jj] = fma(-m1, prj, a1[jj]);
                        }
                    }
                    if (in0) sPerm[pos0] = r0;
                    if (in1) sPerm[pos1] = r1;
#pragma unroll
                    for (int j = 0; j < PW; ++j) {
                        if (in0 && j < nbp) sLU[r0 * ldlu + K0 + j] = a0[j];
                        if (TWO && in1 && j < nbp) sLU[r1 * ldlu + K0 + j] = a1[j];
                    }
                    // (B) the panel's rows of U right of it: column c of the pivot rows R_1 .. R_PW-1 takes the eliminations of the
                    // panel's earlier columns, in their order.  The multipliers -- pivot row j's entries of the panel's columns
                    // before j -- are read where the loop above has just put them (one address for the wavefront: a broadcast)
                    int Rr[PW];
#pragma unroll
                    for (int j = 0; j < PW; ++j)
                        Rr[j] = ((TWO && (pvl[j] >> 6)) ? __builtin_amdgcn_readlane(r1, pvl[j] & 63) : __builtin_amdgcn_readlane(r0, pvl[j] & 63)) * ldlu;
                    if constexpr (PW == 4) {
                        auto of_pivot = [&](int j_, R v0, R v1) -> R {
                            return (TWO && (pvl[j_] >> 6)) ? lane_get(v1, pvl[j_] & 63) : lane_get(v0, pvl[j_] & 63);
                        };
                        R lmq[6];      // l10, l20, l21, l30, l31, l32: pivot row j's multipliers of the columns before it
                        lmq[0] = of_pivot(1, a0[0], TWO ? a1[0] : (R)0.0); lmq[1] = of_pivot(2, a0[0], TWO ? a1[0] : (R)0.0); lmq[2] = of_pivot(2, a0[1], TWO ? a1[1] : (R)0.0);
                        lmq[3] = of_pivot(3, a0[0], TWO ? a1[0] : (R)0.0); lmq[4] = of_pivot(3, a0[1], TWO ? a1[1] : (R)0.0); lmq[5] = of_pivot(3, a0[2], TWO ? a1[2] : (R)0.0);
                        for (int c = base + lane; c < m; c += 64) {
                            const R u0 = sLU[Rr[0] + c];
                            if (nbp > 1) {
                                const R u1 = fma(-lmq[0], u0, sLU[Rr[1] + c]);
                                sLU[Rr[1] + c] = u1;
                                if (nbp > 2) {
                                    const R u2 = fma(-lmq[2], u1, fma(-lmq[1], u0, sLU[Rr[2] + c]));
                                    sLU[Rr[2] + c] = u2;
                                    if (nbp > 3) sLU[Rr[3] + c] = fma(-lmq[5], u2, fma(-lmq[4], u1, fma(-lmq[3], u0, sLU[Rr[3] + c])));
                                }
                            }
                        }
                    } else {
                        for (int c = base + lane; c < m; c += 64) {
                            R u[PW];
#pragma unroll
                            for (int j = 0; j < PW; ++j) u[j] = sLU[Rr[j < nbp ? j : 0] + c];
#pragma unroll
                            for (int j = 1; j < PW; ++j) {
                                if (j >= nbp) break;
                                R t_ = u[j];
#pragma unroll
                                for (int i = 0; i < j; ++i) t_ = fma(-sLU[Rr[j] + K0 + i], u[i], t_);
                                u[j] = t_;
                                sLU[Rr[j] + c] = t_;
                                asm volatile("" ::: "memory");      // (row j + 1's multipliers are not requested before row j is done)
                            }
                        }
                    }
                };
                if constexpr (pw == 4) {
                    if ((m - K0) > 64) panel(std::true_type{}, std::integral_constant<int, 4>{});
                    else panel(std::false_type{}, std::integral_constant<int, 4>{});
                } else {
                    panel(std::false_type{}, std::integral_constant<int, 8>{});
                }
            }
#ifdef DPILQR_PHASE_STAMPS
            bph[5] += __builtin_amdgcn_s_memtime() - lu_t0;      // (slot 5, "S5": the panels, wave 0's clock)
#endif
            __syncthreads();
            // (C) the trailing block: positions and columns from `base` on, 16 x 16 tiles dealt to the wavefronts; an eight-column
            // panel is two chained products, reduction rows K0 .. K0 + 3 first
            if (base < m) {
                const int tb = base >> 4, nt = ((m + 15) >> 4) - tb;
                for (int tl = wave; tl < nt * nt; tl += kBigThreads / 64) {
                    const int it = tb + tl / nt, jt = tb + tl - (tl / nt) * nt;
                    const int pos_a = 16 * it + c16, col = 16 * jt + c16;
                    const bool red = g16 < nbp, red2 = g16 + 4 < nbp;
                    const bool va = pos_a >= base && pos_a < m, vb = col >= base && col < m;
                    const int ra = sPerm[min(pos_a, mk - 1)] * ldlu + K0;
                    const R a = (red && va) ? -sLU[ra + g16] : (R)0.0;
                    const R b = (red && vb) ? sLU[sPerm[K0 + (red ? g16 : 0)] * ldlu + min(col, mk - 1)] : (R)0.0;
                    R a2 = (R)0.0, b2 = (R)0.0;
                    if constexpr (pw == 8) {
                        a2 = (red2 && va) ? -sLU[ra + (red2 ? g16 + 4 : 0)] : (R)0.0;
                        b2 = (red2 && vb) ? sLU[sPerm[K0 + (red2 ? g16 + 4 : 0)] * ldlu + min(col, mk - 1)] : (R)0.0;
                    }
                    acc_t cc;
                    int rw[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int pos = 16 * it + Mfma<R>::row(v, g16);
                        rw[v] = sPerm[min(pos, mk - 1)] * ldlu + min(col, mk - 1);
                        cc[v] = sLU[rw[v]];
                    }
                    cc = Mfma<R>::mac(a, b, cc);
                    if constexpr (pw == 8) cc = Mfma<R>::mac(a2, b2, cc);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int pos = 16 * it + Mfma<R>::row(v, g16);
                        if (pos >= base && pos < m && col >= base && col < m) sLU[rw[v]] = cc[v];
                    }
                }
            }
            __syncthreads();
        };
        {
            int K0 = 0;
#ifdef DPILQR_BIG_LU_PANEL4
            for (; K0 < m; K0 += 4) lu_step(K0, std::integral_constant<int, 4>{});
#else
            // (eight wide with two rows per lane too: 107 spilled registers)
            for (; K0 < m && (m - K0) > 64; K0 += 4) lu_step(K0, std::integral_constant<int, 4>{});
            for (; K0 < m; K0 += 8) lu_step(K0, std::integral_constant<int, 8>{});
#endif
        }
#elif !defined(DPILQR_BIG_LU_PLAIN)
        // With LOOK-AHEAD (round 3): while fifteen wavefronts apply column kk's eliminations to the columns from kk + 2 on, the
        // first wavefront applies them to column kk + 1 alone and then searches THAT column's pivot, exchanges and inverts --
        // the search, the exchange and the fp64 division were a third of every column's critical path (238 k of a step's 1.47 M
        // clocks in the plain form below, -DDPILQR_BIG_LU_PLAIN for A/B, which also costs two workgroup barriers per column
        // instead of one).  Same pivots, same multipliers, every entry updated by the same operations in the same order.  The
        // permutation is double-buffered: a column's eliminations read perm_kk while perm_kk+1 is being written (both exchanges
        // that separate the two buffers are applied by the one lane that decides them).  (Measured and dropped: panels of eight
        // columns factorised by one wavefront, 40 instead of 160 barriers -- 290 k: the per-column chain through LDS, not
        // the barriers, sets the pace.  Round 5: a BLOCKED right-looking LU -- sixteen-column panels factorised by one wavefront
        // entirely in registers (lane = row, wave-wide arg-max, pivot row through v_readlane, the exchange two position
        // numbers changing owners), U_12 one thread per column, the trailing block one thread per row and four columns, 15
        // barriers instead of 80, factors / pivots / permutation bit for bit those of this form (fp64 and fp32) -- measured
        // SLOWER, 217 k against 172 k: a panel column is ~370 instructions of one wavefront that has its SIMD to itself.  Dropped.)
        // (Also round 5: the look-ahead wavefront keeps the arg-max of the entries it has just formed AND every lane divides for its own
        // candidate while the arg-max runs -- below; without the early division the same arrangement bought nothing, 32.0 against
        // 32.5 ms per pass of one item, with it 29.6 -> 28.9.  Measured and dropped, bit-identical: the other wavefronts' updates in
        // groups of 3 x 3 entries with every kind of operand requested before any is used, 32.4 against 32.5.  A column is a chain of
        // about a dozen dependent LDS round trips, a division and a barrier.)
        int* sPermB = sPiv;                               // the second permutation buffer (sPiv is otherwise unused)
        for (int r = tid; r < mk; r += kBigThreads) sPermB[r] = r;
        if (wave == 0) {                                  // column 0's pivot, before the loop
            R best = -1.0;
            int piv = 0;
            for (int ps = (lane & 15); ps < m; ps += 16) {
                const R v = fabs(sLU[ps * ldlu]);
                if (v > best) { best = v; piv = ps; }
            }
#define DPILQR_ARGMAX_STEP(CTRL)                                                                \
            {                                                                               \
                const R ob = dpp_val<CTRL>(best);                                           \
                const int op = dpp_i32<CTRL>(piv);                                          \
                if (ob > best || (ob == best && op < piv)) { best = ob; piv = op; }         \
            }
            DPILQR_ARGMAX_STEP(0xB1) DPILQR_ARGMAX_STEP(0x4E) DPILQR_ARGMAX_STEP(0x141) DPILQR_ARGMAX_STEP(0x140)
            if (lane == 0) {
                sPerm[0] = piv; sPerm[piv] = 0;             // perm_0 lives in sPerm (even columns), perm_1 will live in sPermB
                const R pv = sLU[piv * ldlu];
                if (!(best > (R)0.0)) sFlag[0] = 1;        // zero (or NaN) pivot: np.linalg.solve would raise
                sInv[0] = (pv == (R)0.0) ? (R)0.0 : (R)1.0 / pv;
                sFlag[1] = piv;                            // the exchange (0, piv), to be replayed into the other buffer
            }
        }
        __syncthreads();
        for (int kk = 0; kk < m; ++kk) {
            const int* pk = (kk & 1) ? sPermB : sPerm;     // perm_kk
            int* pn = (kk & 1) ? sPerm : sPermB;           // becomes perm_kk+1
            const R inv = sInv[kk];
            const R* prow = sLU + pk[kk] * ldlu;
            if (wave == 0) {
                if (kk + 1 < m) {
                    // Column kk + 1, all rows below the pivot, its pivot AND the pivot's reciprocal in one pass (round 5): a lane keeps
                    // the largest of the entries it has just formed -- magnitude, position, row -- and divides for ITS candidate
                    // while the arg-max runs (the division, ~40 dependent instructions, used to follow the search on one lane);
                    // the winner's reciprocal and row arrive with the arg-max, nothing is read back, and what the deciding lane
                    // needs of the permutation is requested before the pass.  pn holds perm_kk-1; replaying the exchange (kk, b0)
                    // that made perm_kk gives it pk's entries there, so they are read from pk -- early -- and the lane only
                    // writes.  Same pivots (first position of the largest magnitude), same entries, the same reciprocal 1 / pivot.
                    const int b0 = sFlag[1];
                    const int pk_a0 = pk[kk], pk_b0 = pk[b0], rk = pk[kk + 1];
                    R best = -1.0, bval = 0.0;
                    int piv = kk + 1, brow = rk;
                    for (int ps = kk + 1 + lane; ps < m; ps += 64) {
                        const int r = pk[ps];
                        R* row = sLU + r * ldlu;
                        const R l = row[kk] * inv;
                        const R v = fma(-l, prow[kk + 1], row[kk + 1]);
                        row[kk + 1] = v;
                        const R av = fabs(v);
                        if (av > best) { best = av; piv = ps; bval = v; brow = r; }
                    }
                    R binv = (bval == (R)0.0) ? (R)0.0 : (R)1.0 / bval;      // this lane's candidate's (lanes without one: 0)
#define DPILQR_ARGMAX4(OB, OP, OV, OR_)                                                                  \
                    if ((OB) > best || ((OB) == best && (OP) < piv)) { best = (OB); piv = (OP); binv = (OV); brow = (OR_); }
#define DPILQR_ARGMAX4_DPP(CTRL)                                                                        \
                    {                                                                                   \
                        const R ob = dpp_val<CTRL>(best), ov = dpp_val<CTRL>(binv);                     \
                        const int op = dpp_i32<CTRL>(piv), orw = dpp_i32<CTRL>(brow);                   \
                        DPILQR_ARGMAX4(ob, op, ov, orw)                                                 \
                    }
#define DPILQR_ARGMAX4_XROW(MASK)                                                                       \
                    {                                                                                   \
                        const int src = lane ^ (MASK);                                                  \
                        const R ob = lane_bcast(best, src), ov = lane_bcast(binv, src);                 \
                        const int op = __builtin_amdgcn_ds_bpermute(src << 2, piv), orw = __builtin_amdgcn_ds_bpermute(src << 2, brow); \
                        DPILQR_ARGMAX4(ob, op, ov, orw)                                                 \
                    }
                    DPILQR_ARGMAX4_DPP(0xB1) DPILQR_ARGMAX4_DPP(0x4E) DPILQR_ARGMAX4_DPP(0x141) DPILQR_ARGMAX4_DPP(0x140)
                    if (m - kk - 1 > 16) { DPILQR_ARGMAX4_XROW(16) }     // (the rows of lanes that had entries at all)
                    if (m - kk - 1 > 32) { DPILQR_ARGMAX4_XROW(32) }
#undef DPILQR_ARGMAX4_XROW
#undef DPILQR_ARGMAX4_DPP
#undef DPILQR_ARGMAX4
                    if (lane == 0) {
                        pn[kk] = pk_a0; pn[b0] = pk_b0;          // perm_kk-1 -> perm_kk (the replayed exchange)
                        pn[kk + 1] = brow; pn[piv] = rk;         // ... -> perm_kk+1
                        if (!(best > (R)0.0)) {
                            // a column without a positive finite maximum -- singular, or poisoned -- is flagged; its "pivot" is read back
                            sFlag[0] = 1;
                            const R pv = sLU[brow * ldlu + kk + 1];
                            binv = (pv == (R)0.0) ? (R)0.0 : (R)1.0 / pv;
                        }
                        sInv[kk + 1] = binv;
                        sFlag[1] = piv;
                    }
                }
            } else {
                // the columns from kk + 2 on: positions > kk, a 30 x 32 thread tile over the other fifteen wavefronts
                const int t = tid - 64;
                for (int ps = kk + 1 + (t >> 5); ps < m; ps += (kBigThreads - 64) / 32) {
                    R* row = sLU + pk[ps] * ldlu;
                    const R l = row[kk] * inv;
                    for (int c = kk + 2 + (t & 31); c < m; c += 32) row[c] = fma(-l, prow[c], row[c]);
                }
            }
            __syncthreads();
        }
#undef DPILQR_ARGMAX_STEP
        // the final permutation perm_m-1 into sPerm, where the substitution reads it
        if (((m - 1) & 1) && tid < mk) sPerm[tid] = sPermB[tid];
        __syncthreads();
#else
        for (int kk = 0; kk < m; ++kk) {
            if (wave == 0) {
                // sixteen lanes scan the column, five candidates each at n_u = 80; ties go to the lower position
                R best = -1.0;
                int piv = kk;
                for (int ps = kk + (lane & 15); ps < m; ps += 16) {
                    const R v = fabs(sLU[sPerm[ps] * ldlu + kk]);
                    if (v > best) { best = v; piv = ps; }
                }
#define DPILQR_ARGMAX_STEP(CTRL)                                                                \
                {                                                                               \
                    const R ob = dpp_val<CTRL>(best);                                           \
                    const int op = dpp_i32<CTRL>(piv);                                          \
                    if (ob > best || (ob == best && op < piv)) { best = ob; piv = op; }         \
                }
                DPILQR_ARGMAX_STEP(0xB1) DPILQR_ARGMAX_STEP(0x4E) DPILQR_ARGMAX_STEP(0x141) DPILQR_ARGMAX_STEP(0x140)
#undef DPILQR_ARGMAX_STEP
                if (lane == 0) {
                    const int rk = sPerm[kk], rp = sPerm[piv];
                    sPerm[kk] = rp; sPerm[piv] = rk;
                    const R pv = sLU[rp * ldlu + kk];
                    if (!(best > (R)0.0)) sFlag[0] = 1;            // zero (or NaN) pivot: np.linalg.solve would raise
                    sInv[kk] = (pv == (R)0.0) ? (R)0.0 : (R)1.0 / pv;
                }
            }
            __syncthreads();
            const R inv = sInv[kk];
            const R* prow = sLU + sPerm[kk] * ldlu;
            // trailing update over positions > kk, a 32 x 32 thread tile
            for (int ps = kk + 1 + (tid >> 5); ps < m; ps += kBigThreads / 32) {
                R* row = sLU + sPerm[ps] * ldlu;
                const R l = row[kk] * inv;
                for (int c = kk + 1 + (tid & 31); c < m; c += 32) row[c] = fma(-l, prow[c], row[c]);
            }
            __syncthreads();
        }
#endif
#if defined(DPILQR_BIG_LU_PLAIN) || defined(DPILQR_BIG_LU_COLUMN)
        // multipliers l = a / pivot in place (exactly the values the elimination used)
        for (int e = tid; e < m * 32; e += kBigThreads) {
            const int ps = e >> 5;
            R* row = sLU + sPerm[ps] * ldlu;
            for (int c = (e & 31); c < ps; c += 32) row[c] *= sInv[c];
        }
#endif
        }
        __syncthreads();
        BPHASE(2)

        // ---- S3b: the blocked substitution (riccati_big_solve.inc).  With the team (round 6, twelve-state fp64): the factors and
        // the permutation go to the scratch, one word tells the helpers, and the sixteen column tiles are dealt over ALL parts -- two
        // wavefronts each, a SIMD to itself, instead of sixteen wavefronts sharing four SIMDs (the phase is bound by the issue of
        // the in-tile pivot steps, 95 k clocks of a step's 360 k); every part's arrival then says "all of [K | d] is in place".
        const bool team_solve = kTeamSolve && coop && lu_ok;
        if (team_solve) {
            int tid_c = threadIdx.x;
            asm volatile("" : "+v"(tid_c));
            for (int e = tid_c; e < mk * ldlu; e += kBigThreads) gLU[e] = sLU[e];
            if (tid_c < mk) gLUperm[tid_c] = sPerm[tid_c];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) big_publish(&team->flag_lu, T - t, local);
        }
        {
            const int part_ = 0, nparts_ = team_solve ? nparts : 1;
#include "riccati_big_solve.inc"
        }
        if (team_solve) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) big_arrive(&team->doneK, local);
            if (!big_wait_ge(&team->doneK, (T - 1 - t) * nparts, &sFlag[2], spin_log2)) { gave_up(); return; }
        } else {
            __syncthreads();
        }
        BPHASE(3)

#ifdef DPILQR_BIG_S5_SEPARATE
        {
            const int part_ = 0, nparts_ = 1;
#include "riccati_big_s4.inc"
        }
        __syncthreads();
        BPHASE(4)
#endif

#ifndef DPILQR_BIG_S5_SEPARATE
        {
        if (nparts > 1 && t == T - 1) {     // the team or alone: decided once, when the first step's gains are in place
            if (threadIdx.x == 0) {
                const int joined = __hip_atomic_load(&team->joined, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                // every helper on this workgroup's XCD: hand-overs through its L2 (big_arrive).  (team_dbg & 4: agent scope regardless, A/B)
                const bool one_xcd = big_ld(&team->xccs) == (1 << big_xcc()) && !(team_dbg & 4);
                const int mode = (joined == nparts - 1) ? (one_xcd ? 4 : 1) : 2;
                __hip_atomic_store(&team->mode, mode, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                sFlag[3] = mode;
            }
            __syncthreads();
            coop = sFlag[3] != 2;
            local = sFlag[3] == 4;
        }
        if (coop) {      // [K|d], [Q_ux|Q_u], Q_xx, Q_uu of this step are in the scratch: every thread's stores done, then the word
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) big_publish(&team->flag_k, T - t, local);
        }
        {   // S4 (round 6: the team's too -- one more hand-over, every part's tiles of T3^T stored before any part's S5 reads them)
            const int part_ = 0, nparts_ = coop ? nparts : 1;
#include "riccati_big_s4.inc"
        }
        if (coop) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) big_arrive(&team->done4, local);
            if (!big_wait_ge(&team->done4, (T - t) * nparts, &sFlag[2], spin_log2)) { gave_up(); return; }
        } else {
            __syncthreads();
        }
        BPHASE(4)
        {
            const int part_ = 0, nparts_ = coop ? nparts : 1;
#include "riccati_big_pairs.inc"
        }
        if (coop) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) big_arrive(&team->done, local);
            if (!big_wait_ge(&team->done, (T - t) * nparts, &sFlag[2], spin_log2)) { gave_up(); return; }
        }
        __syncthreads();
        {   // (opaque lane id: hoisted out of the horizon loop this loop's per-lane address was spilled)
            int tid_v = threadIdx.x;
            asm volatile("" : "+v"(tid_v));
            for (int i = tid_v; i < n; i += kBigThreads) sp[i] = gV[(int64_t)i * ldw + n];
        }
        }
#else
        // ---- S5: V = ((Q_xx + T3 [K|d]) + [K|d]^T [Q_ux|Q_u]) + ([K|d]^T [Q_ux|Q_u])^T   rows < n, columns <= n
        {
            const int ti_n = (n + 15) / 16, tj_n = (n1 + 15) / 16;
            for (int job = wave; job < ti_n * tj_n; job += kBigThreads / 64) {
                const int it = job / tj_n, jt = job - it * tj_n;
                acc_t a1 = acc_t{0, 0, 0, 0}, a2 = acc_t{0, 0, 0, 0}, a2t = acc_t{0, 0, 0, 0};
                const int64_t xo = (int64_t)g16 * ldw + 16 * it + c16, yo = (int64_t)g16 * ldw + 16 * jt + c16;
                for (int ks = 0; ks < mk; ks += 4) {
                    const int64_t ro = (int64_t)ks * ldw;
                    const R t3i = gT3[ro + xo], kdi = gKd[ro + xo], gi = gG[ro + xo];
                    const R kdj = gKd[ro + yo], gj = gG[ro + yo];
                    a1 = Mfma<R>::mac(t3i, kdj, a1);
                    a2 = Mfma<R>::mac(kdi, gj, a2);
                    a2t = Mfma<R>::mac(gi, kdj, a2t);
                }
                const int col = 16 * jt + c16;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * it + Mfma<R>::row(v, g16);
                    if (row < n && col <= n) {
                        R* pv = gV + (int64_t)row * ldw + col;
                        *pv = ((*pv + a1[v]) + a2[v]) + a2t[v];
                    }
                }
            }
        }
        __syncthreads();
        BPHASE(5)

        // ---- S6: P <- (V + V^T) / 2 ; p <- V[:, n]
        for (int e = tid; e < n * n; e += kBigThreads) {
            const int i = e / n, j = e - i * n;
            gP[(int64_t)i * ldw + j] = (R)0.5 * (gV[(int64_t)i * ldw + j] + gV[(int64_t)j * ldw + i]);
        }
        for (int i = tid; i < n; i += kBigThreads) sp[i] = gV[(int64_t)i * ldw + n];
#endif
        __syncthreads();
        BPHASE(6)
    }
#ifdef DPILQR_PHASE_STAMPS
    if (tid == 0 && slot == 0) {
        printf("k_riccati_big phases (s_memtime shader-clock ticks per step, ~2.4 GHz): stage %.0f  S1 %.0f  LU %.0f  solve %.0f  S4 %.0f  S5 %.0f  S6 %.0f\n",
               (double)bph[0] / T, (double)bph[1] / T, (double)bph[2] / T, (double)bph[3] / T, (double)bph[4] / T, (double)bph[5] / T,
               (double)bph[6] / T);
    }
#endif
    if (singular && tid == 0 && sFlag[0]) singular[b] = 1;
}

}  // namespace dpilqr
