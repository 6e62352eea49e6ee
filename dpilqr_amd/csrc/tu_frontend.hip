// tu_frontend.hip -- C ABI of the device-side dispatch front / back end (frontend.hpp): dpilqr_dispatch_*.
#include <hip/hip_runtime.h>

#include <cstring>

#include "dpilqr_hip.h"
#include "frontend.hpp"
#include "launch.hpp"

using namespace dpilqr;

static_assert(sizeof(dpilqr_bucket_results) == sizeof(BucketResults), "dpilqr_bucket_results mirrors BucketResults");
static_assert(DPILQR_MAX_AGENTS == kFrontMaxAgents, "one bit per agent");

namespace {
hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
BucketResults to_internal(const dpilqr_bucket_results* R) {
    BucketResults B;
    memcpy(&B, R, sizeof(B));
    return B;
}
}  // namespace

extern "C" {

int32_t dpilqr_dispatch_graph(int32_t S, int32_t N, int32_t k, int32_t n_s, const double* X, const double* radius,
                              int64_t radius_stride, const int32_t* ignore, uint64_t* bits, int32_t* rep, int32_t* size,
                              int32_t* order, int32_t* slot, int32_t* bucket_start, int32_t* bucket_count, void* stream) {
    if (S < 0 || N < 1 || k < 1 || k > kFrontMaxAgents || n_s < 2 || !X || !radius || !bits || !rep || !size || !order || !slot ||
        !bucket_start || !bucket_count)
        return fail(DPILQR_EINVAL, "dispatch_graph: bad argument (k <= %d)", kFrontMaxAgents);
    hipStream_t st = as_stream(stream);
    if (S == 0) {
        HIP_TRY(hipMemsetAsync(bucket_start, 0, sizeof(int32_t) * (k + 1), st));
        HIP_TRY(hipMemsetAsync(bucket_count, 0, sizeof(int32_t) * (k + 1), st));
        return DPILQR_OK;
    }
    const int64_t n = (int64_t)S * k;
    const dim3 grid((unsigned)((n + 127) / 128)), block(128);
    {   // (k + 1) x kSortThreads counters: above 64 KiB at k = 64 -- ask for it before anything of this call is queued
        const int32_t rc_lds = allow_lds(k_bucket_sort, sizeof(int32_t) * (k + 1) * kSortThreads);
        if (rc_lds) return rc_lds;
    }
    hipLaunchKernelGGL(k_graph_bits, grid, block, 0, st, S, N, k, n_s, X, radius, radius_stride,
                       reinterpret_cast<unsigned long long*>(bits));
    hipLaunchKernelGGL(k_dedup, grid, block, 0, st, S, k, reinterpret_cast<const unsigned long long*>(bits), ignore, rep, size);
    hipLaunchKernelGGL(k_bucket_sort, dim3(1), dim3(kSortThreads), sizeof(int32_t) * (k + 1) * kSortThreads, st, S, k, rep, size,
                       order, bucket_start, bucket_count, slot);
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t dpilqr_dispatch_gather(int32_t k, int32_t n_s, int32_t n_c, int32_t T, int32_t n_rows, int32_t kc, const int32_t* order,
                               int32_t first, int32_t count, const uint64_t* bits, const double* X, const double* U,
                               const double* xf, int64_t xf_stride, double* x0_out, double* xf_out, double* U_out,
                               int32_t* members, void* stream) {
    if (k < 1 || k > kFrontMaxAgents || kc < 1 || kc > k || count < 0 || first < 0 || !order || !bits || !X || !U || !xf || !x0_out ||
        !xf_out || !U_out)
        return fail(DPILQR_EINVAL, "dispatch_gather: bad argument");
    if (count == 0) return DPILQR_OK;
    hipLaunchKernelGGL(k_gather_bucket, dim3(count), dim3(128), 0, as_stream(stream), k, n_s, n_c, T, n_rows, kc, order, first, count,
                       reinterpret_cast<const unsigned long long*>(bits), X, U, xf, xf_stride, x0_out, xf_out, U_out, members);
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t dpilqr_dispatch_gather_params(int32_t count, int32_t kc, int32_t width, int32_t elem_bytes, const int32_t* members,
                                      const void* src, void* out, void* stream) {
    if (count < 0 || kc < 1 || width < 1 || !members || !src || !out || (elem_bytes != 4 && elem_bytes != 8))
        return fail(DPILQR_EINVAL, "dispatch_gather_params: bad argument");
    const int64_t n = (int64_t)count * kc * width;
    if (n == 0) return DPILQR_OK;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (elem_bytes == 8)
        hipLaunchKernelGGL((k_gather_params<double>), grid, block, 0, as_stream(stream), count, kc, width, members,
                           static_cast<const double*>(src), static_cast<double*>(out));
    else
        hipLaunchKernelGGL((k_gather_params<int32_t>), grid, block, 0, as_stream(stream), count, kc, width, members,
                           static_cast<const int32_t*>(src), static_cast<int32_t*>(out));
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t dpilqr_dispatch_stitch(int32_t S, int32_t k, int32_t n_s, int32_t n_c, int32_t T, const uint64_t* bits, const int32_t* rep,
                               const int32_t* size, const int32_t* slot, const dpilqr_bucket_results* results, double* X_dec,
                               double* U_dec, void* stream) {
    if (S < 0 || k < 1 || k > kFrontMaxAgents || !bits || !rep || !size || !slot || !results || !X_dec || !U_dec)
        return fail(DPILQR_EINVAL, "dispatch_stitch: bad argument");
    if (S == 0) return DPILQR_OK;
    hipLaunchKernelGGL(k_stitch, dim3((unsigned)((int64_t)S * k)), dim3(128), 0, as_stream(stream), S, k, n_s, n_c, T,
                       reinterpret_cast<const unsigned long long*>(bits), rep, size, slot, to_internal(results), X_dec, U_dec);
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t dpilqr_dispatch_pack_rows(int32_t S, int32_t k, int32_t n_s, int32_t n_c, int32_t T, const uint64_t* bits,
                                  const int32_t* rep, const int32_t* size, const int32_t* slot,
                                  const dpilqr_bucket_results* results, int32_t* row_of, int32_t* n_rows, double* rows,
                                  int64_t row_len, void* stream) {
    if (S < 0 || k < 1 || k > kFrontMaxAgents || !bits || !rep || !size || !slot || !results || !row_of || !n_rows)
        return fail(DPILQR_EINVAL, "dispatch_pack_rows: bad argument");
    if (rows && row_len < 1 + (int64_t)(T + 1) * n_s + (int64_t)T * n_c) return fail(DPILQR_EINVAL, "dispatch_pack_rows: row_len too small");
    hipStream_t st = as_stream(stream);
    if (S == 0) { HIP_TRY(hipMemsetAsync(n_rows, 0, sizeof(int32_t), st)); return DPILQR_OK; }
    const BucketResults R = to_internal(results);
    const unsigned long long* b = reinterpret_cast<const unsigned long long*>(bits);
    hipLaunchKernelGGL(k_local_rows, dim3(1), dim3(kSortThreads), 0, st, S, k, b, rep, size, slot, R, row_of, n_rows);
    if (rows)   // rows == NULL: only count (the caller sizes the buffer from n_rows, or knows the bound S * k)
        hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)((int64_t)S * k)), dim3(128), 0, st, S, k, n_s, n_c, T, b, rep, size, slot, R,
                           row_of, rows, row_len);
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t dpilqr_dispatch_scatter_rows(int64_t n_rows_total, int32_t k, int32_t n_s, int32_t n_c, int32_t T, const double* rows,
                                     int64_t row_len, double* X_dec, double* U_dec, void* stream) {
    if (n_rows_total < 0 || k < 1 || !rows || !X_dec || !U_dec) return fail(DPILQR_EINVAL, "dispatch_scatter_rows: bad argument");
    if (n_rows_total == 0) return DPILQR_OK;
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)n_rows_total), dim3(128), 0, as_stream(stream), n_rows_total, k, n_s, n_c, T,
                       rows, row_len, X_dec, U_dec);
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t dpilqr_random_setup(int32_t S, int64_t seed0, int32_t k, int32_t n_s, int32_t n_d, double var, double energy,
                            double* x0, double* xf, void* stream) {
    if (S < 0 || k < 1 || k > kFrontMaxAgents || n_d < 1 || n_d > 3 || n_s < n_d || !x0 || !xf || seed0 < 0 ||
        seed0 + (int64_t)S > 0xffffffffll)
        return fail(DPILQR_EINVAL, "random_setup: bad argument (k <= %d, n_d in 1..3, seeds in [0, 2^32))", kFrontMaxAgents);
    if (S == 0) return DPILQR_OK;
    hipLaunchKernelGGL(k_random_setup, dim3((S + 63) / 64), dim3(64), 0, as_stream(stream), S, seed0, k, n_s, n_d, var, energy, x0, xf);
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

}  // extern "C"
