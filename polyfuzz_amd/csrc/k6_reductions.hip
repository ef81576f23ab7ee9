// K6 -- the two reductions PolyFuzz runs on the hot path's output:
//   * precision_recall_curve (reference polyfuzz/metrics.py:12-53): for thresholds p_k, how many
//     similarities are >= p_k and their mean -- a histogram over the threshold bins + suffix sums;
//   * single_linkage on a self-match top-1 result (reference polyfuzz/linkage.py:5-53 as called from
//     PolyFuzz._create_groups, polyfuzz.py:459-484) -- see the second half of this file.
// Both are a few hundred KB of HBM-resident data: latency-bound, one small kernel each.
#include "pfz_internal.h"

#include <math.h>

namespace pfz {

constexpr int kMaxThresholds = 4096;

// bin of v = number of thresholds <= v (thresholds ascending); sums in int64 fixed point (value * scale,
// rounded to nearest): integer sums are order-independent, so the curve is bit-reproducible.
__global__ __launch_bounds__(1024) void k_pr_hist(const double *__restrict__ sim, int64_t n, const double *__restrict__ thr,
                                                   int32_t n_thr, double scale, unsigned long long *__restrict__ cnt_out,
                                                   long long *__restrict__ sum_out)
{
    __shared__ double s_thr[kMaxThresholds];
    __shared__ unsigned long long s_cnt[kMaxThresholds + 1];
    __shared__ long long s_sum[kMaxThresholds + 1];
    for (int t = threadIdx.x; t < n_thr; t += 1024) s_thr[t] = thr[t];
    for (int t = threadIdx.x; t <= n_thr; t += 1024) {
        s_cnt[t] = 0ull;
        s_sum[t] = 0ll;
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) {
        const double v = sim[i];
        if (!(v == v)) continue;                 // NaN is never >= a threshold
        int lo = 0, hi = n_thr;                  // first threshold > v
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_thr[mid] <= v) lo = mid + 1;
            else hi = mid;
        }
        if (lo == 0) continue;
        atomicAdd(&s_cnt[lo], 1ull);
        atomicAdd((unsigned long long *)&s_sum[lo], (unsigned long long)llrint(v * scale));
    }
    __syncthreads();
    for (int t = threadIdx.x; t <= n_thr; t += 1024) {
        if (s_cnt[t]) {
            atomicAdd(&cnt_out[t], s_cnt[t]);
            atomicAdd((unsigned long long *)&sum_out[t], (unsigned long long)s_sum[t]);
        }
    }
}

}  // namespace pfz

using namespace pfz;

extern "C" {

int pfz_pr_curve_host(pfz_ctx *ctx, const double *sim, int64_t n, const double *thresholds, int32_t n_thr,
                      int64_t *count_ge, double *sum_ge)
{
    PFZ_REQUIRE(ctx && thresholds && count_ge && sum_ge && (n == 0 || sim), "pfz_pr_curve_host: NULL argument");
    PFZ_REQUIRE(n >= 0 && n_thr >= 1 && n_thr <= kMaxThresholds, "pfz_pr_curve_host: 1 <= n_thresholds <= %d", kMaxThresholds);
    for (int32_t k = 1; k < n_thr; ++k)
        PFZ_REQUIRE(thresholds[k] >= thresholds[k - 1], "pfz_pr_curve_host: thresholds must ascend");
    double max_abs = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double a = fabs(sim[i]);
        if (a == a && a > max_abs) max_abs = a;
    }
    PFZ_REQUIRE(max_abs < 1e300, "pfz_pr_curve_host: similarities must be finite");
    // fixed-point scale: n * max|v| * scale < 2^62
    int e = 0;
    frexp((double)(n > 0 ? n : 1) * (max_abs > 0.0 ? max_abs : 1.0), &e);
    const double scale = ldexp(1.0, 61 - e);
    PFZ_HIP(hipSetDevice(ctx->device));
    struct Buf {
        void *p = nullptr;
        ~Buf() { if (p) pool_free(p); }
    } d_sim, d_thr, d_cnt, d_sum;
    PFZ_TRY(pool_alloc(ctx, &d_sim.p, (size_t)(n > 0 ? n : 1) * sizeof(double)));
    PFZ_TRY(pool_alloc(ctx, &d_thr.p, (size_t)n_thr * sizeof(double)));
    PFZ_TRY(pool_alloc(ctx, &d_cnt.p, (size_t)(n_thr + 1) * sizeof(uint64_t)));
    PFZ_TRY(pool_alloc(ctx, &d_sum.p, (size_t)(n_thr + 1) * sizeof(int64_t)));
    PFZ_TRY(copy_h2d(ctx, d_sim.p, sim, (size_t)n * sizeof(double)));
    PFZ_TRY(copy_h2d(ctx, d_thr.p, thresholds, (size_t)n_thr * sizeof(double)));
    PFZ_HIP(hipMemsetAsync(d_cnt.p, 0, (size_t)(n_thr + 1) * sizeof(uint64_t), ctx->stream));
    PFZ_HIP(hipMemsetAsync(d_sum.p, 0, (size_t)(n_thr + 1) * sizeof(int64_t), ctx->stream));
    if (n > 0) {
        const unsigned grid = (unsigned)std::min<int64_t>((n + 1023) / 1024, 256);
        hipLaunchKernelGGL(k_pr_hist, dim3(grid), dim3(1024), 0, ctx->stream, (const double *)d_sim.p, n, (const double *)d_thr.p,
                           n_thr, scale, (unsigned long long *)d_cnt.p, (long long *)d_sum.p);
        PFZ_HIP(hipGetLastError());
    }
    std::vector<uint64_t> cnt((size_t)n_thr + 1);
    std::vector<int64_t> sum((size_t)n_thr + 1);
    PFZ_TRY(copy_d2h(ctx, cnt.data(), d_cnt.p, cnt.size() * sizeof(uint64_t)));
    PFZ_TRY(copy_d2h(ctx, sum.data(), d_sum.p, sum.size() * sizeof(int64_t)));
    // bin b holds the values with exactly b thresholds <= v: suffix sums give ">= p_k" (101 values: host)
    uint64_t c = 0;
    int64_t s = 0;
    for (int32_t k = n_thr; k >= 1; --k) {
        c += cnt[(size_t)k];
        s += sum[(size_t)k];
        count_ge[k - 1] = (int64_t)c;
        sum_ge[k - 1] = (double)s / scale;
    }
    return PFZ_OK;
}

}  // extern "C"
