// cost.hpp -- device-side cost evaluation / quadraticisation pieces.
//
// Replaces the reference's Python cost plugins for the recognised types:
//   ReferenceCost  cost.py:79-101   (x-xf)^T Q (x-xf) + u^T R u, no 1/2 (quirk Q8)
//   ProximityCost  cost.py:117-171  sum_pairs min(0, d_ij - r)^2
//   quadraticize_distance cost.py:269-315
//   GameCost       cost.py:197-239  w_prox * prox + w_ref * sum_i ref_i
#pragma once
#include <hip/hip_runtime.h>

#include "dpilqr_hip.h"

namespace dpilqr {

// index of pair (i<j) in itertools.combinations(range(k),2) order (util.py:56, cost.py:143-144)
__host__ __device__ inline int pair_index(int i, int j, int k) { return i * (2 * k - i - 1) / 2 + (j - i - 1); }

// One pair of ProximityCost.__call__ (cost.py:117-133): fmin(0, |a-b| - r)^2 over the first nd coordinates.
// R: the arithmetic type (double; float in the fp32 arm of BASELINE config 5's tolerance study)
template <typename R>
__device__ inline R pair_cost(const R* a, const R* b, int nd, R radius) {
    R s = 0.0;
    for (int c = 0; c < nd; ++c) {
        const R df = a[c] - b[c];
        s += df * df;
    }
    // far pairs contribute exactly 0: skip the square root when d^2 is safely beyond r^2
    // (fmin(0, d - r) = 0 whenever d > r; the margin covers the rounding of r*r and of sqrt)
    if (s > radius * radius * (R(1.0) + R(sizeof(R) == 8 ? 1e-12 : 1e-5))) return R(0.0);
    const R m = fmin(R(0.0), sqrt(s) - radius);
    return m * m;
}

// quadraticize_distance (cost.py:269-315): g[3], H[9]; entries at or beyond nd are zero.
template <typename R>
__device__ inline void pair_quadraticize(const R* pa, const R* pb, int nd, R radius, R* g, R* H) {
    R a[3] = {0.0, 0.0, 0.0}, b[3] = {0.0, 0.0, 0.0};
    for (int c = 0; c < 3; ++c)
        if (c < nd) { a[c] = pa[c]; b[c] = pb[c]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) H[c] = 0.0;
    const R dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    const R dist = sqrt(dx * dx + dy * dy + dz * dz);
    if (dist > radius) return;  // active iff not (distance > radius): quirk Q7
    const R gs = 2 * (dist - radius) / dist;
    const R dd[3] = {dx, dy, dz};
    // the cross terms use the distance recomputed as |a|^2 + |b|^2 - 2 a.b (cost.py:293-303)
    const R h2a = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    const R h2b = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    const R dalt = sqrt((h2a + h2b) - 2 * (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]));
    const R cross = 2 * radius / (dalt * dalt * dalt);
    const R d3 = dist * dist * dist;
    R HH[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) HH[c] = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) HH[c * 3 + c] = 2 * radius * (dd[c] * dd[c]) / d3 - 2 * radius / dist + 2;
    HH[1] = HH[3] = (dx * dy) * cross;
    HH[2] = HH[6] = (dx * dz) * cross;
    HH[5] = HH[7] = (dy * dz) * cross;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (r < nd) g[r] = gs * dd[r];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (r < nd && c < nd) H[r * 3 + c] = HH[r * 3 + c];
    }
}

// ReferenceCost.__call__ (cost.py:79-83) for one agent: ((e @ M) @ e) [+ ((u @ R) @ u)].  The weights and the goal
// come from the descriptor (fp64) and are rounded to the arithmetic type on use.
// W: the weights' and the goal's storage type -- double in the descriptor, R where a kernel has staged them (forward.hpp: LDS)
template <int NS, int NC, typename R, typename W = double>
__device__ inline R ref_cost(const R* x, const R* u, const W* xf, const W* M, const W* Rm, bool terminal) {
    R e[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) e[i] = x[i] - R(xf[i]);
    R c = 0.0;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        R v = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i) v += e[i] * R(M[i * NS + j]);
        c += v * e[j];
    }
    if (terminal) return c;
    R cu = 0.0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        R v = 0.0;
#pragma unroll
        for (int i = 0; i < NC; ++i) v += u[i] * R(Rm[i * NC + j]);
        cu += v * u[j];
    }
    return c + cu;
}

// per-item views of the batch descriptor (batch stride 0 = shared by the whole batch)
struct ItemParams {
    const int32_t* model;
    const int32_t* n_dims;
    const double* xf;
    const double* Q;
    const double* R;
    const double* Qf;
    double radius;
};

__device__ inline ItemParams item_params(const dpilqr_batch_desc& D, int b) {
    ItemParams p;
    p.model = D.model + (int64_t)b * D.model_bstride;
    p.n_dims = D.n_dims + (int64_t)b * D.n_dims_bstride;
    p.xf = D.xf + (int64_t)b * D.xf_bstride;
    p.Q = D.Q + (int64_t)b * D.Q_bstride;
    p.R = D.R + (int64_t)b * D.R_bstride;
    p.Qf = D.Qf + (int64_t)b * D.Qf_bstride;
    p.radius = D.radius[(int64_t)b * D.radius_bstride];
    return p;
}

// ProximityCost.__call__ picks PLANAR distances when every agent has the same n_dims
// (cost.py:122-123 -> util.py:48 default n_d=2, quirk Q5), else min(n_dims_i, n_dims_j).
__device__ inline bool homogeneous_ndims(const int32_t* n_dims, int k) {
    bool h = true;
    for (int i = 1; i < k; ++i) h = h && (n_dims[i] == n_dims[0]);
    return h;
}

}  // namespace dpilqr
