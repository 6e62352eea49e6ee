// frontend.hpp -- the dispatch front and back end of solve_distributed for many scenarios, on the device.
//
// Reference: solve_distributed (distributed.py:25-103) builds an interaction graph (define_inter_graph_threshold
// :224-247), splits the problem into one sub-problem per AGENT -- its closed neighbourhood (util.split_graph :102-117,
// problem.split problem.py:36-47) --, solves them one by one and stitches the owners' columns back (:74-75).  Agents
// with the same neighbourhood get identical sub-problems (quirk Q11).  For S Monte-Carlo scenarios of one k-agent
// problem everything between the trajectories and the sub-problem batches is array work over (S, k):
//
//   k_graph_bits      neighbourhood of every (scenario, agent) as a 64-bit mask           [graph]
//   k_dedup           representative agent of every distinct neighbourhood of a scenario, cluster size   [Q11]
//   k_bucket_sort     stable counting sort of the representatives by cluster size: one contiguous list per size
//   k_gather_bucket   x0 / U0 / x_f (and, for heterogeneous teams, per-agent parameters) of one size's sub-problems
//   k_stitch          the owners' columns of every solved sub-problem -> X_dec, U_dec
//   k_pack_rows / k_scatter_rows   the same through one row per (scenario, agent): what a rank contributes to, and takes
//                     from, the path's single all-gather when the sub-problems are sharded over GPUs
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dpilqr {

constexpr int kFrontMaxAgents = 64;   // one bit per agent

// bits[s][i]: bit j set iff agent j is within 2 * radius (planar) of agent i on any sampled row of X[s]; bit i always set.
// X[S][N][k * n_s]; rows sampled as slice(0, N + 1, max(N // 10, 1)) (distributed.py:229-235).
static __global__ void k_graph_bits(int S, int N, int k, int n_s, const double* __restrict__ X,
                                    const double* __restrict__ radius, int64_t radius_stride,
                                    unsigned long long* __restrict__ bits) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)S * k) return;
    const int s = (int)(idx / k), i = (int)(idx - (int64_t)s * k);
    const double thr = 2 * radius[(int64_t)s * radius_stride];
    const int step = (N / 10 > 1) ? N / 10 : 1;
    const double* Xs = X + (int64_t)s * N * k * n_s;
    unsigned long long m = 1ull << i;
    for (int j = 0; j < k; ++j) {
        if (j == i) continue;
        for (int r = 0; r < N; r += step) {
            const double* row = Xs + (int64_t)r * k * n_s;
            const double dx = row[i * n_s] - row[j * n_s], dy = row[i * n_s + 1] - row[j * n_s + 1];
            if (sqrt(dx * dx + dy * dy) < thr) { m |= 1ull << j; break; }
        }
    }
    bits[idx] = m;
}

// rep[s][i]: the first agent of scenario s with the same neighbourhood as agent i (itself if it is the first);
// size[s][i]: number of agents in that neighbourhood.  `active` (may be null): agents to ignore get rep = -1.
static __global__ void k_dedup(int S, int k, const unsigned long long* __restrict__ bits, const int32_t* __restrict__ ignore,
                               int32_t* __restrict__ rep, int32_t* __restrict__ size) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)S * k) return;
    const int s = (int)(idx / k), i = (int)(idx - (int64_t)s * k);
    const unsigned long long m = bits[idx];
    if (ignore && ignore[i]) { rep[idx] = -1; size[idx] = 0; return; }
    int r = i;
    for (int j = 0; j < i; ++j)
        if (bits[(int64_t)s * k + j] == m && !(ignore && ignore[j])) { r = j; break; }
    rep[idx] = r;
    size[idx] = __popcll(m);
}

// Stable counting sort of the representatives ((s, i) with rep == i) by cluster size, one workgroup:
// order[pos] = s * k + i, bucket_start[c], bucket_count[c] for c = 0..k, slot[s * k + i] = position inside its bucket.
constexpr int kSortThreads = 256;
static __global__ __launch_bounds__(kSortThreads) void k_bucket_sort(int S, int k, const int32_t* __restrict__ rep,
                                                                     const int32_t* __restrict__ size, int32_t* __restrict__ order,
                                                                     int32_t* __restrict__ bucket_start, int32_t* __restrict__ bucket_count,
                                                                     int32_t* __restrict__ slot) {
    extern __shared__ int32_t cnt[];          // [k + 1][kSortThreads]: entries of size c in thread t's chunk
    const int t = threadIdx.x;
    const int64_t n = (int64_t)S * k;
    const int64_t chunk = (n + kSortThreads - 1) / kSortThreads, lo = (int64_t)t * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (int c = 0; c <= k; ++c) cnt[c * kSortThreads + t] = 0;
    for (int64_t e = lo; e < hi; ++e)
        if (rep[e] == (int32_t)(e % k)) cnt[size[e] * kSortThreads + t] += 1;
    __syncthreads();
    if (t == 0) {                               // exclusive scan in (size, thread) order: k * 256 additions
        int32_t run = 0;
        for (int c = 0; c <= k; ++c) {
            bucket_start[c] = run;
            for (int q = 0; q < kSortThreads; ++q) {
                const int32_t v = cnt[c * kSortThreads + q];
                cnt[c * kSortThreads + q] = run;
                run += v;
            }
            bucket_count[c] = run - bucket_start[c];
        }
    }
    __syncthreads();
    for (int64_t e = lo; e < hi; ++e) {
        slot[e] = -1;
        if (rep[e] == (int32_t)(e % k)) {
            const int c = size[e];
            const int32_t pos = cnt[c * kSortThreads + t]++;
            order[pos] = (int32_t)e;
            slot[e] = pos - bucket_start[c];
        }
    }
}

// The sub-problems of one cluster size kc, rows [first, first + count) of its bucket: x0, x_f (kc * n_s each), U0
// (T x kc * n_c), members[kc] (ascending agent ids), from the k-agent arrays of their scenarios.
static __global__ void k_gather_bucket(int k, int n_s, int n_c, int T, int n_rows, int kc, const int32_t* __restrict__ order,
                                       int first, int count, const unsigned long long* __restrict__ bits,
                                       const double* __restrict__ X, const double* __restrict__ U,
                                       const double* __restrict__ xf, int64_t xf_stride, double* __restrict__ x0_out,
                                       double* __restrict__ xf_out, double* __restrict__ U_out, int32_t* __restrict__ members) {
    const int j = blockIdx.x;                 // sub-problem within the slice
    if (j >= count) return;
    const int e = order[first + j], s = e / k;
    const unsigned long long m = bits[e];
    __shared__ int mem[kFrontMaxAgents];
    if (threadIdx.x == 0) {
        int p = 0;
        for (int a = 0; a < k; ++a)
            if ((m >> a) & 1ull) mem[p++] = a;
    }
    __syncthreads();
    const double* Xs = X + (int64_t)s * n_rows * k * n_s;        // row 0 = the scenario's current state
    const double* xfs = xf + (int64_t)s * xf_stride;
    for (int q = threadIdx.x; q < kc * n_s; q += blockDim.x) {
        const int p = q / n_s, c = q - p * n_s;
        x0_out[(int64_t)j * kc * n_s + q] = Xs[mem[p] * n_s + c];
        xf_out[(int64_t)j * kc * n_s + q] = xfs[mem[p] * n_s + c];
    }
    const double* Us = U + (int64_t)s * T * k * n_c;
    for (int q = threadIdx.x; q < T * kc * n_c; q += blockDim.x) {
        const int t = q / (kc * n_c), r = q - t * kc * n_c, p = r / n_c, c = r - p * n_c;
        U_out[(int64_t)j * T * kc * n_c + q] = Us[(int64_t)t * k * n_c + mem[p] * n_c + c];
    }
    if (members)
        for (int p = threadIdx.x; p < kc; p += blockDim.x) members[(int64_t)j * kc + p] = mem[p];
}

// Per-agent parameter arrays of the k-agent problem -> per-item arrays of a bucket (heterogeneous teams only):
// out[j][p][w] = src[members[j][p]][w]
template <typename Tv>
static __global__ void k_gather_params(int count, int kc, int width, const int32_t* __restrict__ members,
                                       const Tv* __restrict__ src, Tv* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)count * kc * width) return;
    const int w = (int)(idx % width);
    const int64_t jp = idx / width;
    out[idx] = src[(int64_t)members[jp] * width + w];
}

struct BucketResults {       // solved trajectories of every cluster size: X[kc] is [count][T+1][kc*n_s], U[kc] [count][T][kc*n_c]
    const double* X[kFrontMaxAgents + 1];
    const double* U[kFrontMaxAgents + 1];
    int32_t first[kFrontMaxAgents + 1];      // the slice of the bucket that was solved here: slots [first, first + count)
    int32_t count[kFrontMaxAgents + 1];
};

// where agent i of scenario s finds its columns: the sub-problem of its representative, at its rank inside the cluster
__device__ __forceinline__ bool owner_lookup(int k, const unsigned long long* bits, const int32_t* rep, const int32_t* size,
                                             const int32_t* slot, int64_t e, int& kc, int& sl, int& pos) {
    const int i = (int)(e % k);
    const int r = rep[e];
    if (r < 0) return false;
    const int64_t er = e - i + r;
    kc = size[er];
    sl = slot[er];
    pos = __popcll(bits[e] & ((1ull << i) - 1ull));
    return true;
}

// X_dec[s][t][i*n_s + c], U_dec[s][t][i*n_c + c] <- the owner's columns (distributed.py:74-75); ignored agents stay zero.
static __global__ void k_stitch(int S, int k, int n_s, int n_c, int T, const unsigned long long* __restrict__ bits,
                                const int32_t* __restrict__ rep, const int32_t* __restrict__ size,
                                const int32_t* __restrict__ slot, BucketResults R, double* __restrict__ X_dec,
                                double* __restrict__ U_dec) {
    const int64_t e = blockIdx.x;             // (s, i)
    if (e >= (int64_t)S * k) return;
    int kc, sl, pos;
    if (!owner_lookup(k, bits, rep, size, slot, e, kc, sl, pos)) return;
    sl -= R.first[kc];
    if (sl < 0 || sl >= R.count[kc]) return;  // solved on another rank
    const int s = (int)(e / k), i = (int)(e % k);
    const double* Xb = R.X[kc] + (int64_t)sl * (T + 1) * kc * n_s;
    const double* Ub = R.U[kc] + (int64_t)sl * T * kc * n_c;
    for (int q = threadIdx.x; q < (T + 1) * n_s; q += blockDim.x) {
        const int t = q / n_s, c = q - t * n_s;
        X_dec[((int64_t)s * (T + 1) + t) * k * n_s + i * n_s + c] = Xb[(int64_t)t * kc * n_s + pos * n_s + c];
    }
    for (int q = threadIdx.x; q < T * n_c; q += blockDim.x) {
        const int t = q / n_c, c = q - t * n_c;
        U_dec[((int64_t)s * T + t) * k * n_c + i * n_c + c] = Ub[(int64_t)t * kc * n_c + pos * n_c + c];
    }
}

// Multi-GPU: one row per (scenario, agent) solved on this rank, [index | X columns | U columns], rows in (s, i) order;
// row_of[e] = position of (s, i)'s row in this rank's block (exclusive scan of "solved here"), n_rows its length.
static __global__ __launch_bounds__(kSortThreads) void k_local_rows(int S, int k, const unsigned long long* __restrict__ bits,
                                                                    const int32_t* __restrict__ rep, const int32_t* __restrict__ size,
                                                                    const int32_t* __restrict__ slot, BucketResults R,
                                                                    int32_t* __restrict__ row_of, int32_t* __restrict__ n_rows) {
    __shared__ int32_t part[kSortThreads];
    const int t = threadIdx.x;
    const int64_t n = (int64_t)S * k;
    const int64_t chunk = (n + kSortThreads - 1) / kSortThreads, lo = (int64_t)t * chunk, hi = lo + chunk < n ? lo + chunk : n;
    auto here = [&](int64_t e) {
        int kc, sl, pos;
        if (!owner_lookup(k, bits, rep, size, slot, e, kc, sl, pos)) return false;
        sl -= R.first[kc];
        return sl >= 0 && sl < R.count[kc];
    };
    int32_t c = 0;
    for (int64_t e = lo; e < hi; ++e) c += here(e) ? 1 : 0;
    part[t] = c;
    __syncthreads();
    if (t == 0) {
        int32_t run = 0;
        for (int q = 0; q < kSortThreads; ++q) { const int32_t v = part[q]; part[q] = run; run += v; }
        *n_rows = run;
    }
    __syncthreads();
    int32_t pos = part[t];
    for (int64_t e = lo; e < hi; ++e) row_of[e] = here(e) ? pos++ : -1;
}

static __global__ void k_pack_rows(int S, int k, int n_s, int n_c, int T, const unsigned long long* __restrict__ bits,
                                   const int32_t* __restrict__ rep, const int32_t* __restrict__ size,
                                   const int32_t* __restrict__ slot, BucketResults R, const int32_t* __restrict__ row_of,
                                   double* __restrict__ rows, int64_t row_len) {
    const int64_t e = blockIdx.x;
    if (e >= (int64_t)S * k || row_of[e] < 0) return;
    int kc, sl, pos;
    owner_lookup(k, bits, rep, size, slot, e, kc, sl, pos);
    sl -= R.first[kc];
    double* row = rows + (int64_t)row_of[e] * row_len;
    const double* Xb = R.X[kc] + (int64_t)sl * (T + 1) * kc * n_s;
    const double* Ub = R.U[kc] + (int64_t)sl * T * kc * n_c;
    if (threadIdx.x == 0) row[0] = (double)e;
    for (int q = threadIdx.x; q < (T + 1) * n_s; q += blockDim.x) {
        const int t = q / n_s, c = q - t * n_s;
        row[1 + q] = Xb[(int64_t)t * kc * n_s + pos * n_s + c];
    }
    for (int q = threadIdx.x; q < T * n_c; q += blockDim.x) {
        const int t = q / n_c, c = q - t * n_c;
        row[1 + (T + 1) * n_s + q] = Ub[(int64_t)t * kc * n_c + pos * n_c + c];
    }
}

// gathered rows of all ranks -> X_dec, U_dec (rows whose index is negative are padding)
static __global__ void k_scatter_rows(int64_t n_rows_total, int k, int n_s, int n_c, int T, const double* __restrict__ rows,
                                      int64_t row_len, double* __restrict__ X_dec, double* __restrict__ U_dec) {
    const int64_t r = blockIdx.x;
    if (r >= n_rows_total) return;
    const double* row = rows + r * row_len;
    const double idx = row[0];
    if (!(idx >= 0.0)) return;
    const int64_t e = (int64_t)idx;
    const int64_t s = e / k;
    const int i = (int)(e % k);
    for (int q = threadIdx.x; q < (T + 1) * n_s; q += blockDim.x) {
        const int t = q / n_s, c = q - t * n_s;
        X_dec[(s * (T + 1) + t) * k * n_s + i * n_s + c] = row[1 + q];
    }
    for (int q = threadIdx.x; q < T * n_c; q += blockDim.x) {
        const int t = q / n_c, c = q - t * n_c;
        U_dec[(s * T + t) * k * n_c + i * n_c + c] = row[1 + (T + 1) * n_s + q];
    }
}


// ---- scenario generation ("next" row f4): util.random_setup (util.py:165-195) with random=True, is_rotation=False,
// as scripts/analysis.py:45-54 calls it after np.random.seed(s) -- reproduced bit for bit on the device, one thread per
// scenario: NumPy's legacy global generator is MT19937 seeded by init_genrand(s); np.random.uniform(-1, 1, (k, n_d))
// draws k * n_d doubles in C order, each from two 32-bit outputs, (a >> 5) * 2^26 + (b >> 6) over 2^53, as
// -1 + 2 u; starts first, then goals (util.py:181-182).  normalize_energy (util.py:203-217): subtract the mean position,
// scale so that the summed distance from the origin is `energy` -- with NumPy's summation orders: the mean over agents is
// sequential, the sum of the k norms is NumPy's pairwise sum (eight running partial sums once k >= 8).
struct Mt19937 {
    uint32_t mt[624];
    int idx;
    __device__ void seed(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    __device__ void twist() {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    __device__ uint32_t next32() {
        if (idx >= 624) twist();
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    __device__ double next_double() {
        const uint32_t a = next32() >> 5, b = next32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

__device__ inline double numpy_pairwise_sum(const double* a, int n) {   // numpy/core/src/umath/loops_utils.h, n <= 128
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

static __global__ void k_random_setup(int S, int64_t seed0, int k, int n_s, int n_d, double var, double energy,
                                      double* __restrict__ x0, double* __restrict__ xf) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    Mt19937 g;
    g.seed((uint32_t)(seed0 + s));
    double pos[kFrontMaxAgents * 3], nrm[kFrontMaxAgents];
    for (int which = 0; which < 2; ++which) {
        double* out = (which == 0 ? x0 : xf) + (int64_t)s * k * n_s;
        for (int e = 0; e < k * n_d; ++e) pos[e] = var * (-1.0 + 2.0 * g.next_double());
        if (energy != 0.0) {
            double centre[3];
            for (int c = 0; c < n_d; ++c) {
                double acc = 0.0;
                for (int a = 0; a < k; ++a) acc += pos[a * n_d + c];
                centre[c] = acc / k;
            }
            for (int e = 0; e < k * n_d; ++e) pos[e] -= centre[e % n_d];
            for (int a = 0; a < k; ++a) {
                double q = 0.0;
                for (int c = 0; c < n_d; ++c) q += pos[a * n_d + c] * pos[a * n_d + c];
                nrm[a] = sqrt(q);
            }
            const double scale = energy / numpy_pairwise_sum(nrm, k);
            for (int e = 0; e < k * n_d; ++e) pos[e] *= scale;
        }
        for (int a = 0; a < k; ++a)
            for (int c = 0; c < n_s; ++c) out[a * n_s + c] = c < n_d ? pos[a * n_d + c] : 0.0;
    }
}

}  // namespace dpilqr
