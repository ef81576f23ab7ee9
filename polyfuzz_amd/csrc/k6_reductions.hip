// K6 -- the two reductions PolyFuzz runs on the hot path's output:
//   * precision_recall_curve (reference polyfuzz/metrics.py:12-53): for thresholds p_k, how many
//     similarities are >= p_k and their mean -- a histogram over the threshold bins + suffix sums;
//   * single_linkage on a self-match top-1 result (reference polyfuzz/linkage.py:5-53 as called from
//     PolyFuzz._create_groups, polyfuzz.py:459-484) -- see the second half of this file.
// Both are a few hundred KB of HBM-resident data: latency-bound, one small kernel each.
#include "pfz_internal.h"

#include <math.h>
#include <string.h>

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


// ---------------------------------------------------------------------------
// single_linkage on a self-match top-1 result
// ---------------------------------------------------------------------------
// Reference polyfuzz/linkage.py:28-45, fed by PolyFuzz._create_groups with `model.match(strings)` of UNIQUE
// strings (polyfuzz.py:468-475): row i is (From = strings[i], To = strings[g_i], Similarity); rows with
// Similarity > min_similarity are walked in order,
//     if not mapping.get(From):                       # unmapped, or mapped to cluster 0 (falsy)
//         if not mapping.get(To):  mapping[To] = mapping[From] = cluster_id; cluster_id += 1     ("founding")
//         else:                    mapping[From] = mapping[To]                                    ("adoption")
// The walk is greedy and order-dependent -- union-find / connected components give other clusters -- but with one
// row per string it has a closed structure that a parallel pass can reproduce exactly:
//   * the first kept row t0 founds cluster 0, and because 0 is falsy its two strings count as unmapped from
//     then on: the rest of the walk is the same walk over R = kept rows \ {t0} with ids starting at 1;
//   * string x is mapped before time t  <=>  its own row ran (x in R, x < t)  or an ACTIVE row s < t points to it
//     (g_s = x), where a row is active when its From was still unmapped:  active[x] <=> no active s < x with
//     g_s = x.  That recurrence only looks at smaller indices, so iterating it from "all of R active" fixes
//     index after index (the rounds needed = the longest chain of such forward edges, a handful in practice);
//   * with fin[x] = first active row pointing to x:  row s founds a cluster  <=>  active[s], fin[g_s] == s and
//     g_s was not mapped by its own row (g_s not in R or g_s > s);  cluster ids = 1 + rank of the founding
//     rows in time;  every other active row adopts:  M[s] = M[g_s], resolved by pointer jumping (the chains of
//     active rows strictly decrease in index and end in a founded string);
//   * Python dict insertion order (it orders the reference's `clusters` lists and picks the representative of
//     `cluster_name_map`): a founded To enters at time (s, 0), a From at (its row, 1); t0's two strings keep
//     their first position even when a later founding re-assigns them.
// One workgroup: the data is n ints, every pass is a few microseconds, and the passes need a barrier between them.
constexpr int kLinkT = 1024;
constexpr int32_t kInf = 0x7fffffff;

__device__ inline bool kept_row(int32_t g, float v, int32_t n, double thr)
{
    const double r3 = rint((double)v * 1000.0) / 1000.0;      // np.round(float64(score), 3), the frame's Similarity
    return g >= 0 && g < n && r3 >= 0.001 && r3 > thr;        // (< 0.001 -> To = None, Similarity = 0: never > thr >= 0)
}

__global__ __launch_bounds__(kLinkT) void k_linkage_top1(const int32_t *__restrict__ idx, const float *__restrict__ val,
                                                         int32_t stride, int32_t n, double thr, int32_t *__restrict__ state,
                                                         int32_t *__restrict__ fin, int32_t *__restrict__ ptr,
                                                         int32_t *__restrict__ scan, int32_t *__restrict__ cluster,
                                                         int32_t *__restrict__ key, int32_t *__restrict__ info)
{
    __shared__ int s_t0, s_changed, s_part[kLinkT];
    const int tid = threadIdx.x;
    if (tid == 0) s_t0 = kInf;
    __syncthreads();
    for (int i = tid; i < n; i += kLinkT) {
        const bool f = kept_row(idx[(int64_t)i * stride], val[(int64_t)i * stride], n, thr);
        state[i] = f ? 1 : 0;
        cluster[i] = -1;
        key[i] = -1;
        if (f) atomicMin(&s_t0, i);
    }
    __syncthreads();
    const int t0 = s_t0;
    if (t0 == kInf) {
        if (tid == 0) info[0] = -1, info[1] = 0;
        return;
    }
    // state: bit 1 = row in R, bit 0 = active (current guess)
    for (int i = tid; i < n; i += kLinkT) state[i] = (state[i] && i != t0) ? 3 : 0;
    int rounds = 0;
    for (;;) {
        for (int i = tid; i < n; i += kLinkT) fin[i] = kInf;
        if (tid == 0) s_changed = 0;
        __syncthreads();
        for (int s = tid; s < n; s += kLinkT)
            if (state[s] & 1) atomicMin(&fin[idx[(int64_t)s * stride]], s);
        __syncthreads();
        for (int x = tid; x < n; x += kLinkT) {
            const int st = state[x];
            if (st & 2) {
                const int a = fin[x] > x ? 1 : 0;
                if (a != (st & 1)) {
                    state[x] = 2 | a;
                    s_changed = 1;
                }
            }
        }
        __syncthreads();
        ++rounds;
        const int changed = s_changed;
        __syncthreads();
        if (!changed) break;
    }
    // founding rows (0/1 -> exclusive scan = rank in time) and the adoption pointers
    const int per = (n + kLinkT - 1) / kLinkT;
    const int lo = min(n, tid * per), hi = min(n, lo + per);
    int mine = 0;
    for (int s = lo; s < hi; ++s) {
        const int st = state[s];
        int founding = 0;
        if (st & 1) {
            const int g = idx[(int64_t)s * stride];
            founding = fin[g] == s && !((state[g] & 2) && g < s);
        }
        scan[s] = founding;
        mine += founding;
        ptr[s] = (st & 1) ? idx[(int64_t)s * stride] : s;
    }
    s_part[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < kLinkT; ++t) {
            const int c = s_part[t];
            s_part[t] = run;
            run += c;
        }
        info[2] = run + 1;      // clusters founded (cluster 0 included)
    }
    __syncthreads();
    {
        int run = s_part[tid];
        for (int s = lo; s < hi; ++s) {
            const int f = scan[s];
            scan[s] = run;
            run += f;
        }
    }
    __syncthreads();
    // founded strings: x has an active in-edge and was not mapped by its own row before it
    for (int x = tid; x < n; x += kLinkT) {
        const int f = fin[x];
        if (f != kInf && !((state[x] & 2) && x < f)) {
            cluster[x] = 1 + scan[f];
            key[x] = 2 * f;
        }
    }
    // pointer jumping along the adoption chains
    for (;;) {
        if (tid == 0) s_changed = 0;
        __syncthreads();
        for (int x = tid; x < n; x += kLinkT) {
            const int p = ptr[x], pp = ptr[p];
            if (pp != p) {
                ptr[x] = pp;
                s_changed = 1;
            }
        }
        __syncthreads();
        const int changed = s_changed;
        __syncthreads();
        if (!changed) break;
    }
    for (int x = tid; x < n; x += kLinkT)
        if (state[x] & 1) {
            cluster[x] = cluster[ptr[x]];
            key[x] = 2 * x + 1;
        }
    __syncthreads();
    if (tid == 0) {     // cluster 0: the first kept row's To and From, wherever they ended up they entered the dict first
        const int g0 = idx[(int64_t)t0 * stride];
        if (cluster[g0] < 0) cluster[g0] = 0;
        if (cluster[t0] < 0) cluster[t0] = 0;
        key[g0] = 2 * t0;
        key[t0] = 2 * t0 + 1;
        info[0] = t0;
        info[1] = rounds;
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

int pfz_linkage_top1(pfz_ctx *ctx, const pfz_topn *result, double min_similarity, int32_t *out_cluster, int32_t *out_key,
                     int32_t *out_info3)
{
    PFZ_REQUIRE(ctx && result && out_cluster && out_key, "pfz_linkage_top1: NULL argument");
    PFZ_REQUIRE(min_similarity >= 0.0, "pfz_linkage_top1: min_similarity must be >= 0 (got %g)", min_similarity);
    PFZ_REQUIRE(result->n_rows < ((int64_t)1 << 30), "pfz_linkage_top1: too many rows");
    const int32_t n = (int32_t)result->n_rows;
    int32_t info[3] = {-1, 0, 0};
    if (n > 0) {
        PFZ_HIP(hipSetDevice(ctx->device));
        struct Buf {
            int32_t *p = nullptr;
            ~Buf() { if (p) pool_free(p); }
        } work;
        PFZ_TRY(pool_alloc(ctx, &work.p, ((size_t)n * 6 + 8) * sizeof(int32_t)));
        int32_t *state = work.p, *fin = state + n, *ptr = fin + n, *scan = ptr + n, *cluster = scan + n, *key = cluster + n,
                *d_info = key + n;
        hipLaunchKernelGGL(k_linkage_top1, dim3(1), dim3(kLinkT), 0, ctx->stream, result->idx, result->val, result->ntop, n,
                           min_similarity, state, fin, ptr, scan, cluster, key, d_info);
        PFZ_HIP(hipGetLastError());
        PFZ_TRY(copy_d2h(ctx, out_cluster, cluster, (size_t)n * sizeof(int32_t)));
        PFZ_TRY(copy_d2h(ctx, out_key, key, (size_t)n * sizeof(int32_t)));
        PFZ_TRY(copy_d2h(ctx, info, d_info, sizeof(info)));
    }
    if (out_info3) memcpy(out_info3, info, sizeof(info));
    return PFZ_OK;
}

}  // extern "C"
