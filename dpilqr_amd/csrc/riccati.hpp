// riccati.hpp -- K2: the Riccati backward sweep (ilqrSolver._backward_pass, control.py:116-148).
//
// One workgroup (WAVES x 64 lanes) owns one sub-problem for the whole horizon: the value function
// (p, P) never leaves LDS between time steps, each step's tile record is streamed in from HBM once,
// and K[t], d[t] are streamed out once.  Serial in t (the recursion is), parallel over the batch.
//
//     Q_x  = l_x  + A^T p                  Q_u  = l_u  + B^T p
//     Q_xx = l_xx + (A^T P) A
//     Q_uu = l_uu + (B^T (P + mu I)) B     Q_ux = l_ux + (B^T (P + mu I)) A      (mu on P only: quirk Q6)
//     K = -Q_uu^-1 Q_ux                    d = -Q_uu^-1 Q_u      (LU, partial pivoting = LAPACK dgesv,
//                                                                  np.linalg.solve at control.py:141-142;
//                                                                  Q_uu is indefinite in practice)
//     p <- Q_x + (K^T Q_uu) d + K^T Q_u + Q_ux^T d
//     P <- Q_xx + (K^T Q_uu) K + K^T Q_ux + Q_ux^T K ;  P <- (P + P^T)/2
//
// This file holds the size-generic kernel (run-time n_x, n_u; every matrix in LDS).  Matrix products
// associate exactly as the NumPy expressions do (left to right).
#pragma once
#include <hip/hip_runtime.h>

#include "tiles.hpp"

namespace dpilqr {

struct RiccatiLds {
    // offsets in doubles
    int AB, X, Y, T2, Qux, K, Quu, LU, p, Qx, Qu, pn, misc, total;
    __host__ __device__ RiccatiLds(int n, int m) {
        int o = 0;
        AB = o;  o += n * n + n * m;
        X = o;   o += n * n;
        Y = o;   o += n * n;
        T2 = o;  o += m * n;
        Qux = o; o += m * n;
        K = o;   o += m * (n + 1);
        Quu = o; o += m * m;
        LU = o;  o += m * m;
        p = o;   o += n;
        Qx = o;  o += n;
        Qu = o;  o += m;
        pn = o;  o += n;
        misc = o; o += 2;
        total = o;
    }
};

inline size_t riccati_lds_bytes(int n, int m) { return sizeof(double) * (size_t)RiccatiLds(n, m).total; }

static __global__ __launch_bounds__(256) void k_riccati_generic(int B, int T, int n, int m, const double* __restrict__ tiles,
                                  const double* __restrict__ mu_arr, double* __restrict__ Kout,
                                  double* __restrict__ dout, int32_t* __restrict__ singular,
                                  const int32_t* __restrict__ items, const int32_t* __restrict__ n_items,
                                  int gains_by_item) {
    const int slot = blockIdx.x;
    if (n_items && slot >= *n_items) return;
    const int b = items ? items[slot] : slot;
    if (b >= B) return;
    const int64_t gslot = gains_by_item ? b : slot;   // where K, d of this sub-problem go
    const int tid = threadIdx.x, nth = blockDim.x;
    const TileLayout L(n, m);
    const RiccatiLds O(n, m);
    extern __shared__ double lds[];
    double* sAB = lds + O.AB;   // rows [A[l][:] | B[l][:]], leading dimension n+m (as in the record)
    const int ld = n + m;
    const double* sA = sAB;
    const double* sB = sAB + n;
    double* bufX = lds + O.X;
    double* bufY = lds + O.Y;
    double* sT2 = lds + O.T2;
    double* sQux = lds + O.Qux;
    double* sK = lds + O.K;  // m x (n+1): [K | d] (the solve's right-hand sides)
    double* sQuu = lds + O.Quu;
    double* sLU = lds + O.LU;
    double* sp = lds + O.p;
    double* sQx = lds + O.Qx;
    double* sQu = lds + O.Qu;
    double* spn = lds + O.pn;
    int* sflag = reinterpret_cast<int*>(lds + O.misc);

    const double mu = mu_arr[b];
    const double* base = tiles + (int64_t)slot * (T + 1) * L.stride;
    const int n1 = n + 1;

    // terminal condition: p = l_x(T), P = l_xx(T)   (control.py:125-129)
    {
        const double* rec = base + (int64_t)T * L.stride;
        for (int e = tid; e < n * n; e += nth) bufX[e] = rec[L.oLxx + e];
        for (int e = tid; e < n; e += nth) sp[e] = rec[L.oLx + e];
        if (tid == 0) sflag[0] = 0;
    }
    double* P = bufX;   // current value-function Hessian
    double* W = bufY;   // scratch / next P
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        const double* rec = base + (int64_t)t * L.stride;
        // stage A | B (contiguous in the record) -- the only tile parts used more than once
        for (int e = tid; e < n * ld; e += nth) sAB[e] = rec[L.oA + e];
        __syncthreads();

        // (1) W = A^T P ; T2 = B^T (P + mu I) ; Q_x, Q_u
        for (int e = tid; e < n * n; e += nth) {
            const int i = e / n, j = e - i * n;
            double s = 0.0;
            for (int l = 0; l < n; ++l) s = fma(sA[l * ld + i], P[l * n + j], s);
            W[e] = s;
        }
        for (int e = tid; e < m * n; e += nth) {
            const int a = e / n, j = e - a * n;
            double s = 0.0;
            for (int l = 0; l < n; ++l) {
                const double pv = (l == j) ? P[l * n + j] + mu : P[l * n + j];
                s = fma(sB[l * ld + a], pv, s);
            }
            sT2[e] = s;
        }
        for (int i = tid; i < n + m; i += nth) {
            double s = 0.0;
            if (i < n) {
                for (int l = 0; l < n; ++l) s = fma(sA[l * ld + i], sp[l], s);
                sQx[i] = rec[L.oLx + i] + s;
            } else {
                const int a = i - n;
                for (int l = 0; l < n; ++l) s = fma(sB[l * ld + a], sp[l], s);
                sQu[a] = rec[L.oLu + a] + s;
            }
        }
        __syncthreads();

        // (2) Q_xx -> P's buffer (P is dead now) ; Q_ux ; Q_uu
        for (int e = tid; e < n * n; e += nth) {
            const int i = e / n, j = e - i * n;
            double s = 0.0;
            for (int l = 0; l < n; ++l) s = fma(W[i * n + l], sA[l * ld + j], s);
            P[e] = rec[L.oLxx + e] + s;
        }
        for (int e = tid; e < m * n; e += nth) {
            const int a = e / n, j = e - a * n;
            double s = 0.0;
            for (int l = 0; l < n; ++l) s = fma(sT2[a * n + l], sA[l * ld + j], s);
            const double q = rec[L.oLux + a * L.ldUG + j] + s;
            sQux[e] = q;
            sK[a * n1 + j] = q;
        }
        for (int e = tid; e < m * m; e += nth) {
            const int a = e / m, c = e - a * m;
            double s = 0.0;
            for (int l = 0; l < n; ++l) s = fma(sT2[a * n + l], sB[l * ld + c], s);
            const double q = rec[L.oLuu + a * L.ldUG + c] + s;
            sQuu[e] = q;
            sLU[e] = q;
        }
        for (int a = tid; a < m; a += nth) sK[a * n1 + n] = sQu[a];  // Q_u is complete since (1)'s barrier
        __syncthreads();

        // (3) LU with partial pivoting on [Q_uu | Q_ux | Q_u]; thread c owns augmented column c
        for (int kk = 0; kk < m; ++kk) {
            // every thread scans pivot column kk (LDS broadcast reads): first row of max |.|
            int piv = kk;
            double best = fabs(sLU[kk * m + kk]);
            for (int r = kk + 1; r < m; ++r) {
                const double v = fabs(sLU[r * m + kk]);
                if (v > best) { best = v; piv = r; }
            }
            if (best == 0.0) {
                if (tid == 0) sflag[0] = 1;
                best = 1.0;  // keep going with finite garbage; the item is flagged singular
            }
            const double pv = sLU[piv * m + kk];
            const double inv = (pv == 0.0) ? 0.0 : 1.0 / pv;
            for (int c = kk + 1 + tid; c < m + n1; c += nth) {
                double* col;
                int ld;
                if (c < m) { col = sLU + c; ld = m; } else { col = sK + (c - m); ld = n1; }
                const double top = col[piv * ld];
                if (piv != kk) { col[piv * ld] = col[kk * ld]; col[kk * ld] = top; }
                for (int r = kk + 1; r < m; ++r) {
                    const int rs = (r == piv) ? kk : r;  // column kk has not been swapped yet
                    const double l = sLU[rs * m + kk] * inv;
                    col[r * ld] = fma(-l, top, col[r * ld]);
                }
            }
            __syncthreads();
            if (tid == 0 && piv != kk) {
                const double a0 = sLU[kk * m + kk];
                sLU[kk * m + kk] = sLU[piv * m + kk];
                sLU[piv * m + kk] = a0;
            }
            __syncthreads();
        }
        // back substitution, thread j owns right-hand side j ; then K = -X, d = -x
        for (int j = tid; j < n1; j += nth) {
            for (int r = m - 1; r >= 0; --r) {
                double s = sK[r * n1 + j];
                for (int c = r + 1; c < m; ++c) s = fma(-sLU[r * m + c], sK[c * n1 + j], s);
                sK[r * n1 + j] = s / sLU[r * m + r];
            }
            for (int r = 0; r < m; ++r) sK[r * n1 + j] = -sK[r * n1 + j];
        }
        __syncthreads();

        // stream the gains out: K[b][t] (m x n), d[b][t] (m)
        {
            double* Kt = Kout + (gslot * T + t) * m * n;
            double* dt_ = dout + (gslot * T + t) * m;
            for (int e = tid; e < m * n; e += nth) {
                const int a = e / n, j = e - a * n;
                Kt[e] = sK[a * n1 + j];
            }
            for (int a = tid; a < m; a += nth) dt_[a] = sK[a * n1 + n];
        }

        // (4) T3 = K^T Q_uu  (n x m) -> W
        for (int e = tid; e < n * m; e += nth) {
            const int i = e / m, c = e - i * m;
            double s = 0.0;
            for (int a = 0; a < m; ++a) s = fma(sK[a * n1 + i], sQuu[a * m + c], s);
            W[e] = s;
        }
        __syncthreads();

        // (5) p' and V = Q_xx + T3 K + K^T Q_ux + Q_ux^T K (in place over Q_xx)
        for (int i = tid; i < n; i += nth) {
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int c = 0; c < m; ++c) s1 = fma(W[i * m + c], sK[c * n1 + n], s1);
            for (int a = 0; a < m; ++a) s2 = fma(sK[a * n1 + i], sQu[a], s2);
            for (int a = 0; a < m; ++a) s3 = fma(sQux[a * n + i], sK[a * n1 + n], s3);
            spn[i] = ((sQx[i] + s1) + s2) + s3;
        }
        for (int e = tid; e < n * n; e += nth) {
            const int i = e / n, j = e - i * n;
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int c = 0; c < m; ++c) s1 = fma(W[i * m + c], sK[c * n1 + j], s1);
            for (int a = 0; a < m; ++a) s2 = fma(sK[a * n1 + i], sQux[a * n + j], s2);
            for (int a = 0; a < m; ++a) s3 = fma(sQux[a * n + i], sK[a * n1 + j], s3);
            P[e] = ((P[e] + s1) + s2) + s3;
        }
        __syncthreads();

        // (6) P <- (V + V^T)/2 into the other buffer ; p <- p'
        for (int e = tid; e < n * n; e += nth) {
            const int i = e / n, j = e - i * n;
            W[e] = 0.5 * (P[i * n + j] + P[j * n + i]);
        }
        for (int i = tid; i < n; i += nth) sp[i] = spn[i];
        __syncthreads();
        double* tmp = P; P = W; W = tmp;
    }
    if (singular && tid == 0 && sflag[0]) singular[b] = 1;
}

}  // namespace dpilqr
