// K3, two from-rows per wave -- the same sparse cosine top-n as k3_cossim_topn.hip (reference
// polyfuzz/models/_utils.py:82-91, 128-146), bit-identical results, for jobs whose to-side index lives in L2 / Infinity
// Cache (the headline: 100k x 100k), where the row-major kernel is bound by LDS work and instruction issue:
// 0.87 ms of atomics + 1.02 ms of accumulator sweeps against 2.81 ms measured.
//
// What.  Two consecutive from-rows A and B share one wave and ONE accumulator array: every 32-bit LDS word holds two
// 16-bit sums, A's in the lower half, B's in the upper.  The n-grams of both rows sit side by side in the 64 lanes and go
// through ONE owner search per round; a posting of row B adds its value times 65536 (one more DPP instruction per
// step); ONE sweep reads, clears and filters the cells of both rows (v_pk_max_u16).  Per from-row: half the sweep's LDS
// traffic and instructions, half the owner searches.
//
// Exactness.  A 16-bit sum is only a FILTER.  With the 32-bit scale S = 2^k of the exact kernel the 16-bit one is
// S / 2^15, a power of two, so the per-posting product is the exact kernel's divided by 2^15 BEFORE truncation:
// v16 = floor(v32 / 2^15) exactly, and A = sum v16 lies in (E / 2^15 - nnz, E / 2^15] for the exact sum E (nnz = the
// row's n-grams).  If j is in the true top-n and A_n is the n-th largest approximate sum, then A_j >= A_n - nnz
// (otherwise n columns have E > E_j).  So the kernel keeps, per row, every column whose approximate sum is at least
// (the running n-th best approximate sum) - nnz -- the true top-n plus the few columns within nnz / 65536 of the
// threshold, at most 32 -- and k3_pair_finish re-computes the EXACT integer sums of those columns (the same fp32
// products, truncated and added as integers: the exact kernel's arithmetic, order-independent) and selects the top-n by
// the exact keys.  Rows the filter cannot decide go to the exact row-major kernel afterwards (k3_cossim_topn_kernel
// on a row list): more than 32 columns inside the margin (many duplicate to-strings), an n-th best below the margin
// (fewer than n matches worth ~2e-4), more than 64 n-grams.  Results are bit-identical to the row-major kernel's.
#include "k3_core.h"

#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

namespace pfz {

constexpr int kPairCap = 96;             // candidate keys per row (as in the main kernel)
constexpr int kPairKeep = kPairCap - 64;   // ... of which at most this many survive a compaction (and the row)
constexpr int kPairMaxTop = 16;

struct K3PairArgs {
    const int32_t *a_indptr;
    const int32_t *a_idx;
    const float *a_val;
    int32_t n_a;
    const int32_t *tab;
    const int2 *post;
    int32_t nb, n_pieces, ntop, thr0;       // thr0: the exact kernel's strict lower bound, in 32-bit units
    float scale16;
    int32_t exclude_diag;
    int64_t diag_offset;
    int32_t *cand_cols;    // [n_a][32] candidate columns of a row
    int32_t *cand_cnt;     // [n_a] their number; -1: the row goes to the exact kernel
    int32_t *flag_rows;    // [n_a] rows for the exact kernel ...
    int32_t *n_flag;       // ... and how many
    int32_t no_pairs;      // (tests: every row on its own)
    int32_t debug;
};

__device__ inline uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Keep, sorted at cand[0..keep), the ntop best keys and every further key whose sum is within `margin` of the ntop-th best;
// at most kPairKeep (more inside the margin: `over`).  The acceptance threshold follows the ntop-th best.
__device__ inline void compact_margin(uint64_t *cand, TopState &st, int ntop, int margin, int lane, bool &over)
{
    wave_sync();
    constexpr int kPer = (kPairCap + 63) / 64;
    uint64_t e[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int p = lane + 64 * i;
        e[i] = p < st.cnt ? cand[p] : 0ull;
    }
    wave_sync();
    int keep = 0, nth = 0;
    for (int r = 0; r <= kPairKeep; ++r) {
        uint64_t m = e[0];
#pragma unroll
        for (int i = 1; i < kPer; ++i) m = e[i] > m ? e[i] : m;
        const uint64_t best = wave_max_u64(m);
        if (best == 0ull) break;
        const int sum = (int)(uint32_t)(best >> 32);
        if (r >= ntop && sum + margin < nth) break;       // outside the margin: it and everything below it can go
        if (r == kPairKeep) {                             // a 33rd key inside the margin
            over = true;
            break;
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i)
            if (e[i] == best) e[i] = 0ull;
        if (lane == 0) cand[r] = best;
        keep = r + 1;
        if (r == ntop - 1) nth = sum;
    }
    st.cnt = keep;
    if (keep >= ntop) {
        const int t = nth - margin - 1;                   // accept sum > t  <=>  sum >= nth - margin
        st.thr = t > st.thr ? t : st.thr;
    }
    wave_sync();
}

// push the halves of one int4 (columns j0..j0+3) that beat their row's threshold
__device__ inline void push4_pk(uint64_t *candA, uint64_t *candB, TopState &sa, TopState &sb, const int4 &v, int j0, int selfA,
                                int selfB, int ntop, int mA, int mB, int lane, bool &overA, bool &overB)
{
    const uint32_t vv[4] = {(uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j = j0 + c;
        {
            const int w = (int)(vv[c] & 0xffffu);
            const bool pred = w > sa.thr && j != selfA;
            const uint64_t mk = __ballot(pred);
            if (mk) {
                const int pos = sa.cnt + __popcll(mk & ((1ull << lane) - 1ull));
                if (pred) candA[pos] = ((uint64_t)(uint32_t)w << 32) | (uint32_t)(~j);
                sa.cnt += __popcll(mk);
                if (sa.cnt > kPairKeep) compact_margin(candA, sa, ntop, mA, lane, overA);
            }
        }
        {
            const int w = (int)(vv[c] >> 16);
            const bool pred = w > sb.thr && j != selfB;
            const uint64_t mk = __ballot(pred);
            if (mk) {
                const int pos = sb.cnt + __popcll(mk & ((1ull << lane) - 1ull));
                if (pred) candB[pos] = ((uint64_t)(uint32_t)w << 32) | (uint32_t)(~j);
                sb.cnt += __popcll(mk);
                if (sb.cnt > kPairKeep) compact_margin(candB, sb, ntop, mB, lane, overB);
            }
        }
    }
}

// Warm start for both halves: the k-th largest of the lanes' own maxima bounds the k-th largest sum of the block from below
template <int N4>
__device__ inline void warm_pk(const int4 *acc4, int kA, int kB, int lane, int &tA, int &tB)
{
    uint32_t lm = 0u;
#pragma unroll
    for (int t = 0; t < N4 / 128; ++t) {
        const int4 v0 = acc4[t * 128 + lane], v1 = acc4[t * 128 + lane + 64];
        lm = pk_max_u16(pk_max_u16(lm, pk_max_u16((uint32_t)v0.x, (uint32_t)v0.y)), pk_max_u16((uint32_t)v0.z, (uint32_t)v0.w));
        lm = pk_max_u16(pk_max_u16(lm, pk_max_u16((uint32_t)v1.x, (uint32_t)v1.y)), pk_max_u16((uint32_t)v1.z, (uint32_t)v1.w));
    }
    auto kth = [&](int x, int k) {
        int best = 0;
        for (int r = 0; r < k; ++r) {
            best = x;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const int o = __shfl_xor(best, d, 64);
                best = o > best ? o : best;
            }
            if (best == 0) break;
            if (x == best) x = 0;
        }
        return best;
    };
    tA = kth((int)(lm & 0xffffu), kA);
    tB = kth((int)(lm >> 16), kB);
}

template <int C>
__global__ __launch_bounds__(64) void k3_pair_kernel(const K3PairArgs a)
{
    __shared__ __attribute__((aligned(16))) struct {
        int acc[C];
        uint64_t candA[kPairCap];
        uint64_t candB[kPairCap];
    } sm;
    int *const acc = sm.acc;
    uint64_t *const candA = sm.candA, *const candB = sm.candB;
    int *const mark = (int *)(sm.candB + kPairCap) - 64;      // scatter scratch: the tail of row B's candidate buffer
    if ((uint32_t)(uintptr_t)sm.acc != 0u) __builtin_trap();  // layout assumption of run_steps()
    const int lane = threadIdx.x;
    int4 *acc4 = (int4 *)acc;
    constexpr int N4 = C / 4;
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    for (int t = lane; t < C / 4; t += 64) acc4[t] = make_int4(0, 0, 0, 0);
    wave_sync();
    const char *post_bytes = (const char *)a.post;
    const int src4 = (4 * (lane & 15) + (lane >> 4)) * 4;
    const int sub8 = (lane & 15) * 8;
    const int dummy_addr = a.n_pieces << 7;
    const int nb = a.nb, ntop = a.ntop;
    const int n_units = (a.n_a + 1) >> 1;

    // unit = rows (2u, 2u + 1); they run together when their n-grams fit the 64 lanes, else one after the other
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        const int r_first = 2 * u;
        const int q0 = a.a_indptr[r_first], q1 = a.a_indptr[min(r_first + 1, a.n_a)], q2 = a.a_indptr[min(r_first + 2, a.n_a)];
        const bool together = (q2 - q0) <= 64 && !a.no_pairs;
        for (int pass = 0; pass < (together ? 1 : 2); ++pass) {
            // rows of this pass: A (always), B (only when together)
            const int rowA = together ? r_first : r_first + pass;
            if (rowA >= a.n_a) break;
            const int rowB = together && r_first + 1 < a.n_a ? r_first + 1 : -1;
            const int pA0 = together ? q0 : (pass == 0 ? q0 : q1), pA1 = together ? q1 : (pass == 0 ? q1 : q2);
            const int nnzA = pA1 - pA0, nnzB = rowB >= 0 ? q2 - q1 : 0;
            bool overA = nnzA > 64, overB = false;      // (more than 64 n-grams: the exact kernel's slow loop does that row)
            const int nnz = overA ? 0 : nnzA + nnzB;    // lanes in use: [0, nnzA) row A, [nnzA, nnz) row B
            const int64_t sA = (int64_t)rowA + a.diag_offset, sB = (int64_t)rowB + a.diag_offset;
            const int selfA = (a.exclude_diag && sA >= 0 && sA < 0x7fffffff) ? (int)sA : -1;
            const int selfB = (a.exclude_diag && rowB >= 0 && sB >= 0 && sB < 0x7fffffff) ? (int)sB : -1;
            // the filter's margin: a row's approximate sum is below the exact one (in 16-bit units) by less than its n-grams
            const int mA = nnzA, mB = nnzB;
            // accept A > thr.  From the exact bound thr0: E > thr0  =>  A > (thr0 >> 15) - nnz - 1 (may be negative: then
            // cells that sum to 0 could be candidates, which the filter cannot see -- see the end of the row)
            const int lbA = (a.thr0 >> 15) - mA - 1, lbB = (a.thr0 >> 15) - mB - 1;
            TopState sa, sb;
            sa.cnt = sb.cnt = 0;
            sa.pushed = sb.pushed = 0;
            sa.thr = lbA > 0 ? lbA : 0;
            sb.thr = rowB >= 0 ? (lbB > 0 ? lbB : 0) : 0xffff;      // (no row B: nothing passes)

            const int n_blk = nb;
            int b_first = 0;
            if (selfA >= 0 && selfA / C < nb) b_first = selfA / C;
            int cur0 = 0, nxt0 = 0;
            float as0 = 0.f;
            const bool have0 = lane < nnz;
            const int32_t *trow = a.tab;
            if (have0) {
                as0 = a.a_val[pA0 + lane] * a.scale16;
                trow = a.tab + (int64_t)a.a_idx[pA0 + lane] * nb;
                cur0 = trow[b_first];
                nxt0 = trow[b_first + 1];
            }
            bool warmed = false;
            for (int it = 0, b = b_first; it < n_blk && nnz > 0; ++it) {
                const int s = cur0, e = have0 ? nxt0 : cur0;
                const int b_next = b + 1 < nb ? b + 1 : 0;
                const bool touched = __ballot(e > s) != 0;
                if (touched) {
                    if (a.no_pairs == 2) scatter_pieces<false>(acc, post_bytes, mark, e - s, s, as0, lane, src4, sub8, dummy_addr);
                    else scatter_pieces<true>(acc, post_bytes, mark, e - s, s, as0, lane, src4, sub8, dummy_addr, nnzA);
                }
                if (have0 && it + 1 < n_blk) {
                    cur0 = trow[b_next];
                    nxt0 = trow[b_next + 1];
                }
                if (touched) {
                    wave_sync();
                    if (!warmed) {
                        warmed = true;
                        if (ntop <= kWarmMaxTop) {
                            int tA, tB;
                            warm_pk<N4>(acc4, ntop + (selfA >= 0 ? 1 : 0), ntop + (selfB >= 0 ? 1 : 0), lane, tA, tB);
                            // (tX bounds the n-th best approximate sum from below: the filter keeps what is within the margin of it)
                            tA -= mA + 1;
                            tB -= mB + 1;
                            sa.thr = tA > sa.thr ? tA : sa.thr;
                            if (rowB >= 0) sb.thr = tB > sb.thr ? tB : sb.thr;
                        }
                    }
                    const int col0 = b * C;
#pragma unroll 2
                    for (int t = 0; t < N4 / 128; ++t) {
                        const int i0 = t * 128 + lane, i1 = i0 + 64;
                        const int4 v0 = acc4[i0], v1 = acc4[i1];
                        acc4[i0] = make_int4(zero, zero, zero, zero);
                        acc4[i1] = make_int4(zero, zero, zero, zero);
                        const uint32_t mx = pk_max_u16(pk_max_u16(pk_max_u16((uint32_t)v0.x, (uint32_t)v0.y), pk_max_u16((uint32_t)v0.z, (uint32_t)v0.w)),
                                                       pk_max_u16(pk_max_u16((uint32_t)v1.x, (uint32_t)v1.y), pk_max_u16((uint32_t)v1.z, (uint32_t)v1.w)));
                        const uint32_t thr_pk = ((uint32_t)sb.thr << 16) | (uint32_t)sa.thr;
                        if (__ballot(pk_max_u16(mx, thr_pk) != thr_pk)) {          // some half above its row's threshold
                            push4_pk(candA, candB, sa, sb, v0, col0 + i0 * 4, selfA, selfB, ntop, mA, mB, lane, overA, overB);
                            push4_pk(candA, candB, sa, sb, v1, col0 + i1 * 4, selfA, selfB, ntop, mA, mB, lane, overA, overB);
                        }
                    }
                    wave_sync();
                }
                b = b_next;
            }

            // ---- the rows' candidates: what is within the margin of the n-th best, or the exact kernel ----------------
            auto finish = [&](int row, uint64_t *cand, TopState &st, int margin, int lb, bool over, int nnz_row) {
                if (row < 0) return;
                if (nnz_row > 0 && !over) compact_margin(cand, st, ntop, margin, lane, over);
                // the filter cannot see cells that sum to 0: when the exact bound lets them in (lb < 0) they matter as soon as
                // fewer than ntop columns were kept, or the kept ones reach down into the margin of 0
                bool exact = over;
                if (nnz_row > 0 && !over && lb < 0) {
                    const int last = st.cnt >= ntop ? (int)(uint32_t)(cand[ntop - 1] >> 32) : 0;
                    if (st.cnt < ntop || last - margin - 1 < 0) exact = true;
                }
                if (exact) {
                    if (lane == 0) {
                        a.cand_cnt[row] = -1;
                        a.flag_rows[atomicAdd(a.n_flag, 1)] = row;
                    }
                } else {
                    if (lane < st.cnt) a.cand_cols[(int64_t)row * kPairKeep + lane] = (int32_t)(~(uint32_t)cand[lane]);
                    if (lane == 0) a.cand_cnt[row] = nnz_row > 0 ? st.cnt : 0;
                    if (a.debug && st.cnt <= 16 && lane < st.cnt) a.cand_cols[(int64_t)row * kPairKeep + 16 + lane] = (int32_t)(uint32_t)(cand[lane] >> 32);
                }
                wave_sync();
            };
            finish(rowA, candA, sa, mA, lbA, overA, nnzA);
            finish(rowB, candB, sb, mB, lbB, overB, nnzB);
        }
    }
}

// The exact sums of a row's candidate columns and its top-n: one wave per from-row.
__global__ __launch_bounds__(256) void k3_pair_finish(const int32_t *__restrict__ a_indptr, const int32_t *__restrict__ a_idx,
                                                      const float *__restrict__ a_val, int32_t n_a,
                                                      const int32_t *__restrict__ b_indptr, const int32_t *__restrict__ b_idx,
                                                      const float *__restrict__ b_val, const int32_t *__restrict__ cand_cols,
                                                      const int32_t *__restrict__ cand_cnt, int32_t ntop, int32_t thr0, float scale,
                                                      float inv_scale, int32_t *__restrict__ out_idx, float *__restrict__ out_val)
{
    __shared__ int s_k[4][64];
    __shared__ float s_v[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n_a) return;
    const int cnt = cand_cnt[row];
    if (cnt < 0) return;                          // the exact kernel writes this row
    const int p0 = a_indptr[row], nnz = a_indptr[row + 1] - p0;       // (<= 64 here)
    if (lane < nnz) {
        s_k[wave][lane] = a_idx[p0 + lane];
        s_v[wave][lane] = a_val[p0 + lane] * scale;
    }
    wave_sync();
    uint64_t key = 0ull;
    if (lane < cnt) {
        const int j = cand_cols[(int64_t)row * kPairKeep + lane];
        const int e0 = b_indptr[j], e1 = b_indptr[j + 1];
        int sum = 0;
        for (int e = e0; e < e1; ++e) {
            const int kb = b_idx[e];
            int lo = 0, hi = nnz;                 // the from-row's column ids are sorted: binary search
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_k[wave][mid] < kb) lo = mid + 1;
                else hi = mid;
            }
            if (lo < nnz && s_k[wave][lo] == kb) sum += (int)(s_v[wave][lo] * b_val[e]);      // the exact kernel's product, truncated
        }
        if (sum > thr0) key = ((uint64_t)(uint32_t)sum << 32) | (uint32_t)(~j);
    }
    for (int r = 0; r < ntop; ++r) {
        const uint64_t best = wave_max_u64(key);
        if (key == best) key = 0ull;
        if (lane == 0) {
            out_idx[(int64_t)row * ntop + r] = best ? (int32_t)(~(uint32_t)best) : -1;
            out_val[(int64_t)row * ntop + r] = best ? (float)(int32_t)(uint32_t)(best >> 32) * inv_scale : 0.f;
        }
    }
}

static int pair_env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

bool k3_pair_wanted(const pfz_ctx *ctx, const pfz_index *ix, int64_t n_rows, int32_t ntop, int n_slices)
{
    if (!ix->b_indptr || ix->block_cols != 2048 || ntop > kPairMaxTop || n_slices != 1 || ix->n_blocks < 1) return false;
    const int force = pair_env_int("PFZ_K3_PAIR", -1);      // 1: whenever possible (tests), 0: never
    if (force >= 0) return force != 0;
    return n_rows >= 8192;
}

int k3_pair_launch(pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int64_t row_begin, int64_t n_rows, int32_t ntop,
                   int32_t thr0, int scale_log2, int32_t exclude_diag, int64_t diag_offset, pfz_topn *out, int32_t **flag_rows,
                   int32_t **n_flag)
{
    // scratch: [n_flag][cand_cnt n_rows][flag_rows n_rows][cand_cols n_rows x 32]
    const size_t head = 256, cnt_b = ((size_t)n_rows * 4 + 255) & ~(size_t)255;
    PFZ_TRY(ensure_scratch(ctx, head + 2 * cnt_b + (size_t)n_rows * kPairKeep * 4));
    char *base = (char *)ctx->scratch;
    PFZ_HIP(hipMemsetAsync(base, 0, head, ctx->stream));
    K3PairArgs a;
    a.a_indptr = A->indptr + row_begin;
    a.a_idx = A->indices;
    a.a_val = A->data;
    a.n_a = (int32_t)n_rows;
    a.tab = ix->tab;
    a.post = ix->post;
    a.nb = ix->n_blocks;
    a.n_pieces = ix->n_pieces;
    a.ntop = ntop;
    a.thr0 = thr0;
    a.scale16 = (float)ldexp(1.0, scale_log2 - 15);
    a.exclude_diag = exclude_diag;
    a.diag_offset = diag_offset + row_begin;
    a.n_flag = (int32_t *)base;
    a.cand_cnt = (int32_t *)(base + head);
    a.flag_rows = (int32_t *)(base + head + cnt_b);
    a.cand_cols = (int32_t *)(base + head + 2 * cnt_b);
    a.no_pairs = pair_env_int("PFZ_K3_PAIR_SINGLE", 0);
    a.debug = getenv("PFZ_K3_PAIR_DEBUG") ? 1 : 0;
    const int64_t units = (n_rows + 1) / 2;
    const int64_t max_grid = (int64_t)ctx->prop.multiProcessorCount * 16 * 64;
    const unsigned grid = (unsigned)(units < max_grid ? units : max_grid);
    hipLaunchKernelGGL((k3_pair_kernel<2048>), dim3(grid), dim3(64), 0, ctx->stream, a);
    const float scale = (float)ldexp(1.0, scale_log2), inv_scale = (float)ldexp(1.0, -scale_log2);
    hipLaunchKernelGGL(k3_pair_finish, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, ctx->stream, a.a_indptr, a.a_idx, a.a_val,
                       (int32_t)n_rows, ix->b_indptr, ix->b_indices, ix->b_data, a.cand_cols, a.cand_cnt, ntop, thr0, scale, inv_scale,
                       out->idx + row_begin * ntop, out->val + row_begin * ntop);
    PFZ_HIP(hipGetLastError());
    if (getenv("PFZ_K3_PAIR_DEBUG")) {           // (development aid: what the filter kept)
        std::vector<int32_t> cnt((size_t)n_rows), cols((size_t)std::min<int64_t>(n_rows, 4) * kPairKeep);
        int32_t nf = 0;
        PFZ_HIP(hipStreamSynchronize(ctx->stream));
        PFZ_HIP(hipMemcpy(&nf, a.n_flag, 4, hipMemcpyDeviceToHost));
        PFZ_HIP(hipMemcpy(cnt.data(), a.cand_cnt, cnt.size() * 4, hipMemcpyDeviceToHost));
        PFZ_HIP(hipMemcpy(cols.data(), a.cand_cols, cols.size() * 4, hipMemcpyDeviceToHost));
        long sum = 0, neg = 0, zero = 0;
        for (int32_t c : cnt) { if (c < 0) ++neg; else { sum += c; if (!c) ++zero; } }
        fprintf(stderr, "k3_pair: rows %lld flagged %d (cnt<0: %ld) zero %ld mean kept %.2f\n", (long long)n_rows, nf, neg, zero,
                (double)sum / std::max<long>(1, (long)cnt.size() - neg));
        for (int r = 0; r < (int)std::min<int64_t>(n_rows, 4); ++r) {
            fprintf(stderr, "  row %d cnt %d:", r, cnt[r]);
            for (int i = 0; i < std::max(0, std::min(cnt[r], 12)); ++i)
                fprintf(stderr, " %d(%d)", cols[(size_t)r * kPairKeep + i], cols[(size_t)r * kPairKeep + 16 + i]);
            fprintf(stderr, "\n");
        }
    }
    *flag_rows = a.flag_rows;
    *n_flag = a.n_flag;
    return PFZ_OK;
}

}  // namespace pfz
