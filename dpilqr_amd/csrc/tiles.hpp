// tiles.hpp -- K1: device-side tile producer.
//
// For every (item b, time step t) it evaluates what the reference computes per step inside
// ilqrSolver._backward_pass (control.py:125-133):
//     GameCost.quadraticize(X[t],U[t])            cost.py:208-239  (+ :85-101, :135-171, :269-315)
//     MultiDynamicalModel.linearize(X[t],U[t])    dynamics.py:173-186 -> bbdynamics.cpp linearize_*
// and writes ONE dense tile record (layout: TileLayout) to HBM.  The records are what the Riccati
// sweep (riccati.hpp) streams back in.  One 64-lane wavefront per (b,t):
//   phase 1  lanes 0..k-1 linearise "their" agent and form e = x - xf;  lanes stride the i<j pairs
//            and evaluate the pair gradient/Hessian of the proximity penalty   -> LDS
//   phase 2  all lanes assemble the dense record and store it with unit-stride 8-byte stores.
// The record is dense because dense (n_x,n_x) matrices are the reference's plugin contract.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "cost.hpp"
#include "models.hpp"

namespace dpilqr {

struct TileLayout {
    // One record = [ AB n x (n+m) | L_xx n x n | UG m x (m+n) | L_x n | L_u m ], every component starting on a
    // 16-byte boundary.  Two components are row-interleaved because the sweep consumes them as stacked matrices:
    //   AB row l = [ A[l][0..n) | B[l][0..m) ]          (operands of [A|B]^T P and [T1;T2][A|B])
    //   UG row a = [ L_uu[a][0..m) | L_ux[a][0..n) ]    (added onto the rows of [Q_uu | Q_ux])
    int n, m;              // n_x, n_u
    int oA, oB, ldAB;      // A[l][i] at oA + l*ldAB + i ; B[l][a] at oB + l*ldAB + a (oB = oA + n)
    int oLxx;
    int oLuu, oLux, ldUG;  // L_uu[a][c] at oLuu + a*ldUG + c ; L_ux[a][j] at oLux + a*ldUG + j (oLux = oLuu + m)
    int oLx, oLu;
    int stride;            // doubles per record (even -> 16-byte aligned records)
    __host__ __device__ static int even(int x) { return (x + 1) & ~1; }
    __host__ __device__ TileLayout(int n_x, int n_u) : n(n_x), m(n_u) {
        ldAB = n + m;
        oA = 0;
        oB = n;
        oLxx = even(n * ldAB);
        ldUG = m + n;
        oLuu = even(oLxx + n * n);
        oLux = oLuu + m;
        oLx = even(oLuu + m * ldUG);
        oLu = oLx + n;     // [L_x | L_u] contiguous: the sweep reads them as one (n+m)-vector
        stride = even(oLu + m);
    }
};

// LDS doubles of one time step's phase-1 results: per-agent A, B blocks, x - xf, u, per-pair gradient + Hessian
__host__ __device__ inline int make_tiles_lds_doubles(int k, int ns, int nc) {
    return k * ns * ns + k * ns * nc + k * ns + k * nc + (k * (k - 1) / 2) * 12;
}

// SPARSE = true writes only the entries that can be non-zero for ANY item of this shape (the agents' diagonal
// blocks of A, B, L_uu, the diagonal and the position-coupling blocks of L_xx, L_x, L_u): 430 of the 1330
// doubles of a cfg2 record.  It requires a buffer whose structural zeros are already in place -- the solve
// loop zeroes its tile workspace once per call and then reuses the slots for items of the same shape -- and
// leaves in HBM exactly the dense records the sweep reads.  SPARSE = false (the API's default) writes all of it.
// dyn_only (SPARSE only): additionally skip what does not depend on (X, U) -- A, B of a model with linear dynamics
// and L_uu = w_ref (R + R^T) -- when the caller has put those in place once and every item that will use the
// slot shares them (one linear model, one R for the batch).
template <int NS, int NC, bool SPARSE>
__global__ __launch_bounds__(64) void k_make_tiles(dpilqr_batch_desc D, const double* __restrict__ X,
                                                    const double* __restrict__ U, double* __restrict__ tiles,
                                                    const int32_t* __restrict__ items,
                                                    const int32_t* __restrict__ n_items, int ts, int dyn_only) {
    // one wavefront handles `ts` consecutive time steps of one item: phase 1 spreads (step, agent) and
    // (step, pair) over the lanes, phase 2 writes the records one after the other
    const int slot = blockIdx.y;
    if (n_items && slot >= *n_items) return;
    const int b = items ? items[slot] : slot;
    const int k = D.k, T = D.T;
    const int n = k * NS, m = k * NC;
    const int lane = threadIdx.x;
    const int npairs = k * (k - 1) / 2;
    const ItemParams P = item_params(D, b);
    const TileLayout L(n, m);
    const int t_first = blockIdx.x * ts;                 // records t_first .. t_first+ts-1 (clipped to T)
    const int n_t = min(ts, T + 1 - t_first);
    const int per_t = make_tiles_lds_doubles(k, NS, NC);

    extern __shared__ double lds_all[];
    int* sPair = reinterpret_cast<int*>(lds_all + (size_t)ts * per_t);   // [npairs][2] agent indices (i < j)
    for (int p = lane; p < npairs; p += 64) {
        int i = 0, rem = p;
        while (rem >= k - 1 - i) { rem -= k - 1 - i; ++i; }
        sPair[2 * p] = i; sPair[2 * p + 1] = i + 1 + rem;
    }
    __syncthreads();

    // ---- phase 1a: per-(step, agent) linearisation and error vector
    for (int idx = lane; idx < n_t * k; idx += 64) {
        const int tl = idx / k, a = idx - tl * k;
        const int t = t_first + tl;
        const bool terminal = (t == T);
        double* lds = lds_all + (size_t)tl * per_t;
        double* sA = lds;
        double* sB = sA + k * NS * NS;
        double* sE = sB + k * NS * NC;
        double* sU = sE + k * NS;
        const double* xt = X + ((int64_t)b * (T + 1) + t) * n;
        const double* ut = U + ((int64_t)b * T + (terminal ? 0 : t)) * m;
        {
        double x[NS], u[NC], A[NS * NS], Bm[NS * NC];
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = xt[a * NS + i];
#pragma unroll
        for (int i = 0; i < NC; ++i) u[i] = terminal ? 0.0 : ut[a * NC + i];
        if (!terminal && !(SPARSE && dyn_only)) {
            linearize_rt<NS>(P.model[a], x, u, D.dt, A, Bm);
#pragma unroll
            for (int i = 0; i < NS * NS; ++i) sA[a * NS * NS + i] = A[i];
#pragma unroll
            for (int i = 0; i < NS * NC; ++i) sB[a * NS * NC + i] = Bm[i];
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) sE[a * NS + i] = x[i] - P.xf[a * NS + i];
#pragma unroll
        for (int i = 0; i < NC; ++i) sU[a * NC + i] = u[i];
        }
    }
    // ---- phase 1b: pair derivatives, pairs in itertools.combinations order
    for (int idx = lane; idx < n_t * npairs; idx += 64) {
        const int tl = idx / npairs, p = idx - tl * npairs;
        const int i = sPair[2 * p], j = sPair[2 * p + 1];
        double* lds = lds_all + (size_t)tl * per_t;
        double* sG = lds + k * NS * NS + k * NS * NC + k * NS + k * NC;
        double* sH = sG + npairs * 3;
        const double* xt = X + ((int64_t)b * (T + 1) + t_first + tl) * n;
        const int nd = min(P.n_dims[i], P.n_dims[j]);  // cost.py:145
        double g[3], H[9];
        pair_quadraticize(xt + i * NS, xt + j * NS, nd, P.radius, g, H);
#pragma unroll
        for (int c = 0; c < 3; ++c) sG[p * 3 + c] = g[c];
#pragma unroll
        for (int c = 0; c < 9; ++c) sH[p * 9 + c] = H[c];
    }
    __syncthreads();

    const double wr = D.w_ref, wp = D.w_prox;
    for (int tl = 0; tl < n_t; ++tl) {
    const int t = t_first + tl;
    const bool terminal = (t == T);
    double* lds = lds_all + (size_t)tl * per_t;
    double* sA = lds;                      // [k][NS*NS]
    double* sB = sA + k * NS * NS;         // [k][NS*NC]
    double* sE = sB + k * NS * NC;         // [k*NS]  x - xf
    double* sU = sE + k * NS;              // [k*NC]
    double* sG = sU + k * NC;              // [npairs][3]
    double* sH = sG + npairs * 3;          // [npairs][9]
    double* rec = tiles + ((int64_t)slot * (T + 1) + t) * L.stride;  // records are indexed by list position

    if (SPARSE) {
        // ---- phase 2 (sparse): only the structurally non-zero entries; same values as the dense path below
        constexpr int PD = NS < 3 ? NS : 3;    // position sub-block edge of the coupling blocks
        if (!terminal && !dyn_only) {
            for (int e = lane; e < k * NS * NS; e += 64) {
                const int a = e / (NS * NS), r = e - a * NS * NS, li = r / NS, lj = r - li * NS;
                rec[L.oA + (a * NS + li) * L.ldAB + a * NS + lj] = sA[e];
            }
            for (int e = lane; e < k * NS * NC; e += 64) {
                const int a = e / (NS * NC), r = e - a * NS * NC, li = r / NC, lj = r - li * NC;
                rec[L.oB + (a * NS + li) * L.ldAB + a * NC + lj] = sB[e];
            }
            for (int e = lane; e < k * NC * NC; e += 64) {
                const int a = e / (NC * NC), r = e - a * NC * NC, li = r / NC, lj = r - li * NC;
                const double* R = P.R + a * NC * NC;
                rec[L.oLuu + (a * NC + li) * L.ldUG + a * NC + lj] = wr * (R[li * NC + lj] + R[lj * NC + li]);
            }
        }
        if (!terminal) {
            for (int j = lane; j < m; j += 64) {
                const int a = j / NC, lj = j - a * NC;
                const double* R = P.R + a * NC * NC;
                double v = 0.0;
#pragma unroll
                for (int i = 0; i < NC; ++i) v += sU[a * NC + i] * (R[i * NC + lj] + R[lj * NC + i]);
                rec[L.oLu + j] = wr * v;
            }
        }
        for (int e = lane; e < k * NS * NS; e += 64) {   // diagonal blocks of L_xx
            const int a = e / (NS * NS), r = e - a * NS * NS, li = r / NS, lj = r - li * NS;
            const double* M = (terminal ? P.Qf : P.Q) + a * NS * NS;
            double v = wr * (M[li * NS + lj] + M[lj * NS + li]);
            if (k > 1 && li < 3 && lj < 3) {
                double acc = 0.0;
                for (int o = 0; o < k; ++o) {
                    if (o == a) continue;
                    const int p = (o < a) ? pair_index(o, a, k) : pair_index(a, o, k);
                    acc += sH[p * 9 + li * 3 + lj];
                }
                v += wp * acc;
            }
            rec[L.oLxx + (a * NS + li) * n + a * NS + lj] = v;
        }
        for (int e = lane; e < npairs * 2 * PD * PD; e += 64) {   // position-coupling blocks of L_xx, both mirrors
            const int p = e / (2 * PD * PD), r = e - p * 2 * PD * PD, mirror = r / (PD * PD), q = r - mirror * PD * PD;
            const int li = q / PD, lj = q - li * PD;
            const int ai = mirror ? sPair[2 * p + 1] : sPair[2 * p], aj = mirror ? sPair[2 * p] : sPair[2 * p + 1];
            double acc = 0.0;
            acc += -sH[p * 9 + li * 3 + lj];
            rec[L.oLxx + (ai * NS + li) * n + aj * NS + lj] = 0.0 + wp * acc;
        }
        for (int j = lane; j < n; j += 64) {
            const int a = j / NS, lj = j - a * NS;
            const double* M = (terminal ? P.Qf : P.Q) + a * NS * NS;
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) v += sE[a * NS + i] * (M[i * NS + lj] + M[lj * NS + i]);
            v = wr * v;
            if (k > 1 && lj < 3) {
                double acc = 0.0;
                for (int o = 0; o < k; ++o) {
                    if (o == a) continue;
                    if (o < a) acc += -sG[pair_index(o, a, k) * 3 + lj];
                    else       acc += sG[pair_index(a, o, k) * 3 + lj];
                }
                v += wp * acc;
            }
            rec[L.oLx + j] = v;
        }
        continue;
    }
    // ---- phase 2: dense record.  A, B only for t < T (record T never has them read).
    if (!terminal) {
        for (int e = lane; e < n * n; e += 64) {
            const int i = e / n, j = e - i * n;
            const int ai = i / NS, aj = j / NS;
            rec[L.oA + i * L.ldAB + j] = (ai == aj) ? sA[ai * NS * NS + (i - ai * NS) * NS + (j - aj * NS)] : 0.0;
        }
        for (int e = lane; e < n * m; e += 64) {
            const int i = e / m, j = e - i * m;
            const int ai = i / NS, aj = j / NC;
            rec[L.oB + i * L.ldAB + j] = (ai == aj) ? sB[ai * NS * NC + (i - ai * NS) * NC + (j - aj * NC)] : 0.0;
        }
        for (int e = lane; e < m * n; e += 64) rec[L.oLux + (e / n) * L.ldUG + (e % n)] = 0.0;  // L_ux = 0 (cost.py:93,231)
        for (int e = lane; e < m * m; e += 64) {
            const int i = e / m, j = e - i * m;
            const int ai = i / NC, aj = j / NC;
            double v = 0.0;
            if (ai == aj) {
                const double* R = P.R + ai * NC * NC;
                const int li = i - ai * NC, lj = j - aj * NC;
                v = wr * (R[li * NC + lj] + R[lj * NC + li]);  // R + R^T (cost.py:63,92)
            }
            rec[L.oLuu + i * L.ldUG + j] = v;
        }
        for (int j = lane; j < m; j += 64) {
            const int a = j / NC, lj = j - a * NC;
            const double* R = P.R + a * NC * NC;
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < NC; ++i) v += sU[a * NC + i] * (R[i * NC + lj] + R[lj * NC + i]);
            rec[L.oLu + j] = wr * v;
        }
    }
    // L_xx = w_ref * blockdiag(Q+Q^T) + w_prox * sum_pairs(+-H)   (cost.py:228-237, 160-169)
    for (int e = lane; e < n * n; e += 64) {
        const int i = e / n, j = e - i * n;
        const int ai = i / NS, aj = j / NS, li = i - ai * NS, lj = j - aj * NS;
        double v = 0.0;
        if (ai == aj) {
            const double* M = (terminal ? P.Qf : P.Q) + ai * NS * NS;
            v = wr * (M[li * NS + lj] + M[lj * NS + li]);
        }
        if (k > 1 && li < 3 && lj < 3) {
            double acc = 0.0;
            if (ai == aj) {
                for (int o = 0; o < k; ++o) {  // pairs containing ai, in combinations order
                    if (o == ai) continue;
                    const int p = (o < ai) ? pair_index(o, ai, k) : pair_index(ai, o, k);
                    acc += sH[p * 9 + li * 3 + lj];
                }
            } else {
                const int p = (ai < aj) ? pair_index(ai, aj, k) : pair_index(aj, ai, k);
                acc += -sH[p * 9 + li * 3 + lj];
            }
            v += wp * acc;
        }
        rec[L.oLxx + e] = v;
    }
    // L_x = w_ref * e^T (Q+Q^T) + w_prox * sum_pairs(+-g)
    for (int j = lane; j < n; j += 64) {
        const int a = j / NS, lj = j - a * NS;
        const double* M = (terminal ? P.Qf : P.Q) + a * NS * NS;
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i) v += sE[a * NS + i] * (M[i * NS + lj] + M[lj * NS + i]);
        v = wr * v;
        if (k > 1 && lj < 3) {
            double acc = 0.0;
            for (int o = 0; o < k; ++o) {
                if (o == a) continue;
                if (o < a) acc += -sG[pair_index(o, a, k) * 3 + lj];
                else       acc += sG[pair_index(a, o, k) * 3 + lj];
            }
            v += wp * acc;
        }
        rec[L.oLx + j] = v;
    }
    }   // records of this wavefront
}

// steps per wavefront: as many as fit ~48 KB of LDS, at most 8
inline int make_tiles_steps(int k, int ns, int nc) {
    const int per_t = make_tiles_lds_doubles(k, ns, nc);
    static const int cap = route_int("DPILQR_TILES_STEPS", 8);   // tuning knob
    int ts = (48 * 1024 / 8) / per_t;
    return ts < 1 ? 1 : (ts > cap ? cap : ts);
}
inline size_t make_tiles_lds_bytes(int k, int ns, int nc, int ts) {
    const int npairs = k * (k - 1) / 2;
    return sizeof(double) * ((size_t)ts * make_tiles_lds_doubles(k, ns, nc) + npairs + 2);
}

}  // namespace dpilqr
