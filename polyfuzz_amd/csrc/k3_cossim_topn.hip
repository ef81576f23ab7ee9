// K3 -- sparse cosine top-n: C = A * B^T (CSR x CSR^T) fused with the strict
// lower bound, the self-match diagonal removal and the per-row top-n.
//
// Replaces sparse_dot_topn.awesome_cossim_topn as called at reference
// polyfuzz/models/_utils.py:82, plus _utils.py:84-91,128-146.
//
// Data layout in HBM
//   from side : CSR (indptr int32, indices int32 sorted, data fp32)
//   to side   : inverted index.  The to-rows are cut into blocks of kC rows;
//               for n-gram id k and block b the postings (local to-row, fp32
//               value) are post[tab[k*nb+b] .. tab[k*nb+b+1]).  tab is one
//               int32 array of V*nb+1 offsets, so a from-row walks block b of
//               all its n-grams by reading two neighbouring offsets each.
//
// Kernel (one wave == one workgroup == one from-row at a time)
//   for each to-block b (ascending):
//     for each n-gram k of the row (ascending id): the 64 lanes stride over the
//       (k,b) postings and ds_add_f32 a*b into acc[local to-row] (LDS, kC fp32).
//       Within one (k,b) list every to-row is unique, so there are no
//       same-address collisions inside an instruction, and LDS operations of
//       one wave execute in order: every accumulator receives its terms in
//       ascending k -- the Gustavson order of the CPU reference -- and the
//       result is bit-reproducible.
//     sweep: lanes read acc as float4, write zeros back, and push entries with
//       score > threshold into a per-wave candidate buffer (64-bit keys
//       score_bits<<32 | ~col, so an unsigned max is "score desc, col asc").
//       When the buffer fills, the wave keeps its ntop best (rounds of
//       wave-max) and raises the threshold to the ntop-th score.
//   The final compaction leaves the sorted top-n; lanes write (idx, score).
//
// Roofline: the kernel is bound by the LDS scatter-add + sweep and by the
// posting stream out of L2 / Infinity Cache; the algorithmic HBM-side bytes
// are 8 B per multiply-add (one posting) + 8 B per from-nnz + 8 B per result.
#include "pfz_internal.h"

namespace pfz {

constexpr int kC = 2048;    // to-rows per block == fp32 accumulators per wave (8 KiB LDS)
constexpr int kCap = 256;   // candidate keys per wave (2 KiB LDS)
constexpr int kMaxTop = 128;
constexpr int kSlots = 8;    // posting chunks (64 entries each) in flight per wave

// ---------------------------------------------------------------------------
// inverted-index build
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_index_count(const int32_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ indices, int32_t n_rows,
                                                      int32_t nb, int32_t *__restrict__ tab1 /* tab + 1 */)
{
    // one 16-lane group per to-row: rows have ~13 entries
    const int gid = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    if (gid >= n_rows) return;
    const int p0 = indptr[gid], p1 = indptr[gid + 1];
    const int b = gid / kC;
    for (int p = p0 + sub; p < p1; p += 16) atomicAdd(&tab1[(int64_t)indices[p] * nb + b], 1);
}

__global__ __launch_bounds__(256) void k_index_fill(const int32_t *__restrict__ indptr,
                                                     const int32_t *__restrict__ indices,
                                                     const float *__restrict__ data, int32_t n_rows, int32_t nb,
                                                     int32_t *__restrict__ tab1, int2 *__restrict__ post)
{
    const int gid = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    if (gid >= n_rows) return;
    const int p0 = indptr[gid], p1 = indptr[gid + 1];
    const int b = gid / kC;
    const int local = gid - b * kC;
    for (int p = p0 + sub; p < p1; p += 16) {
        // the order inside one (k,b) list is irrelevant to the results: every
        // to-row occurs at most once per list
        int pos = atomicAdd(&tab1[(int64_t)indices[p] * nb + b], 1);
        post[pos] = make_int2(local, __float_as_int(data[p]));
    }
}

// ---------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------
__device__ inline uint64_t wave_max_u64(uint64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, d, 64);
        uint32_t hi = __shfl_xor((uint32_t)(v >> 32), d, 64);
        uint64_t o = ((uint64_t)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

__device__ inline float readlane_f(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

struct TopState {
    int cnt;     // wave-uniform number of keys in cand[]
    float thr;   // accept score > thr
};

// Keep the ntop best of cand[0..cnt) sorted at cand[0..keep).
__device__ inline void compact(uint64_t *cand, TopState &st, int ntop, int lane)
{
    __syncthreads();
    uint64_t e[kCap / 64];
#pragma unroll
    for (int i = 0; i < kCap / 64; ++i) {
        int p = lane + 64 * i;
        e[i] = p < st.cnt ? cand[p] : 0ull;
    }
    __syncthreads();
    const int keep = st.cnt < ntop ? st.cnt : ntop;
    uint64_t best = 0;
    for (int r = 0; r < keep; ++r) {
        uint64_t m = e[0];
#pragma unroll
        for (int i = 1; i < kCap / 64; ++i) m = e[i] > m ? e[i] : m;
        best = wave_max_u64(m);
#pragma unroll
        for (int i = 0; i < kCap / 64; ++i)
            if (e[i] == best) e[i] = 0ull;
        if (lane == 0) cand[r] = best;
    }
    st.cnt = keep;
    if (keep == ntop) {
        // from now on only scores >= the ntop-th best can matter
        uint32_t bits = (uint32_t)(best >> 32);
        float t = __uint_as_float(bits - 1u);
        st.thr = t > st.thr ? t : st.thr;
    }
    __syncthreads();
}

// Scatter every non-empty (k,b) posting list of one from-row chunk into acc.
//   m       : wave-uniform mask of the lanes whose list [s,e) is non-empty
//   s, e, a : per lane -- lane l owns n-gram k_l of the row: its posting range in
//             this block and the row's value for that n-gram
// Lists are taken in ascending lane (= ascending n-gram id) order and cut into
// 64-entry chunks.  U chunks form a round: all U global loads of a round are
// issued before the first is consumed (the kernel is latency-bound otherwise:
// SQ_WAIT_ANY was 76 % of wave cycles with one load in flight), then the chunks
// are applied strictly in order with a plain LDS read-add-write each.  That is
// safe because a to-row occurs at most once per list (no two lanes of one
// instruction share an address) and the LDS executes a wave's operations in
// program order (a later list reads what an earlier one wrote); it is used
// instead of ds_add_f32 because the LDS float atomic retires ~1 lane per clock.
template <int U>
__device__ inline void scatter_lists(float *acc, const int2 *__restrict__ post, uint64_t m, int s, int e, float a,
                                     int lane)
{
    int cur_q = 0, cur_e = 0;   // wave-uniform: next entry / end of the list being cut
    float cur_a = 0.f;
    while (m != 0 || cur_q < cur_e) {
        int2 pe[U];
        float pa[U];
        bool pv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (cur_q >= cur_e && m != 0) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                cur_q = __builtin_amdgcn_readlane(s, src);
                cur_e = __builtin_amdgcn_readlane(e, src);
                cur_a = readlane_f(a, src);
            }
            const int q = cur_q + lane;
            pv[u] = q < cur_e;
            pa[u] = cur_a;
            pe[u] = make_int2(0, 0);
            if (pv[u]) pe[u] = post[q];
            cur_q += 64;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pv[u]) acc[pe[u].x] = acc[pe[u].x] + pa[u] * __int_as_float(pe[u].y);
            __builtin_amdgcn_wave_barrier();   // keep the compiler from reordering LDS accesses across chunks
        }
    }
}

__global__ __launch_bounds__(64) void k3_cossim_topn_kernel(
    const int32_t *__restrict__ a_indptr, const int32_t *__restrict__ a_idx, const float *__restrict__ a_val,
    int32_t n_a, const int32_t *__restrict__ tab, const int2 *__restrict__ post, int32_t nb, int32_t ntop,
    float lower_bound, int32_t exclude_diag, int64_t diag_offset, int32_t *__restrict__ out_idx,
    float *__restrict__ out_val)
{
    __shared__ __attribute__((aligned(16))) float acc[kC];
    __shared__ __attribute__((aligned(16))) uint64_t cand[kCap];
    const int lane = threadIdx.x;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < kC / 256; ++t) *(float4 *)&acc[(t * 64 + lane) * 4] = zero4;
    __syncthreads();

    for (int row = blockIdx.x; row < n_a; row += gridDim.x) {
        const int p0 = a_indptr[row], p1 = a_indptr[row + 1];
        const int nnz = p1 - p0;
        const int64_t self_col64 = (int64_t)row + diag_offset;
        const int self_col = (exclude_diag && self_col64 >= 0 && self_col64 < 0x7fffffff) ? (int)self_col64 : -1;
        TopState st;
        st.cnt = 0;
        st.thr = lower_bound;

        // registers for the first 64 n-grams of the row (covers almost every row);
        // the offset-table entry of the next block is always one block ahead in flight
        int k0 = 0, cur0 = 0, nxt0 = 0;
        float a0 = 0.f;
        const bool have0 = lane < nnz;
        const int32_t *trow = tab;
        if (have0) {
            k0 = a_idx[p0 + lane];
            a0 = a_val[p0 + lane];
            trow = tab + (int64_t)k0 * nb;
            cur0 = trow[0];
            nxt0 = trow[1];
        }

        for (int b = 0; b < nb; ++b) {
            bool touched = false;
            {
                const int s = cur0, e = have0 ? nxt0 : cur0;
                cur0 = e;
                if (have0 && b + 2 <= nb) nxt0 = trow[b + 2];   // prefetch for block b+1 (tab has V*nb+2 slots)
                const uint64_t m = __ballot(e > s);
                touched = m != 0;
                scatter_lists<kSlots>(acc, post, m, s, e, a0, lane);
            }
            for (int c0 = p0 + 64; c0 < p1; c0 += 64) {  // rows with more than 64 n-grams
                int s = 0, e = 0;
                float a = 0.f;
                if (c0 + lane < p1) {
                    const int k = a_idx[c0 + lane];
                    a = a_val[c0 + lane];
                    s = tab[(int64_t)k * nb + b];
                    e = tab[(int64_t)k * nb + b + 1];
                }
                const uint64_t m = __ballot(e > s);
                touched |= m != 0;
                scatter_lists<kSlots>(acc, post, m, s, e, a, lane);
            }
            if (!touched) continue;
            __syncthreads();  // order the LDS adds before the sweep's reads (single wave: no cost)

            const int col0 = b * kC;
#pragma unroll 2
            for (int t = 0; t < kC / 256; ++t) {
                const int e0 = (t * 64 + lane) * 4;
                const float4 v = *(const float4 *)&acc[e0];
                *(float4 *)&acc[e0] = zero4;
                const float thr = st.thr;
                const bool any = (v.x > thr) | (v.y > thr) | (v.z > thr) | (v.w > thr);
                if (__ballot(any)) {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int j = col0 + e0 + c;
                        const bool pred = vv[c] > st.thr && j != self_col;
                        const uint64_t mk = __ballot(pred);
                        if (mk) {
                            const int pos = st.cnt + __popcll(mk & ((1ull << lane) - 1ull));
                            if (pred) cand[pos] = ((uint64_t)__float_as_uint(vv[c]) << 32) | (uint32_t)(~j);
                            st.cnt += __popcll(mk);
                            if (st.cnt > kCap - 64) compact(cand, st, ntop, lane);
                        }
                    }
                }
            }
            __syncthreads();
        }

        compact(cand, st, ntop, lane);
        for (int r = lane; r < ntop; r += 64) {
            int32_t oi = -1;
            float ov = 0.f;
            if (r < st.cnt) {
                const uint64_t key = cand[r];
                oi = (int32_t)(~(uint32_t)key);
                ov = __uint_as_float((uint32_t)(key >> 32));
            }
            out_idx[(int64_t)row * ntop + r] = oi;
            out_val[(int64_t)row * ntop + r] = ov;
        }
        __syncthreads();
    }
}

}  // namespace pfz

using namespace pfz;

extern "C" {

int pfz_index_build(pfz_ctx *ctx, const pfz_csr *B, pfz_index **out)
{
    PFZ_REQUIRE(ctx && B && out, "pfz_index_build: NULL argument");
    PFZ_HIP(hipSetDevice(ctx->device));
    const int64_t nb = (B->n_rows + kC - 1) / kC;
    const int64_t slots = B->n_cols * nb;
    if (slots >= ((int64_t)1 << 31) - 2) {
        set_error("pfz_index_build: vocabulary %lld x %lld to-blocks exceeds the int32 offset table",
                  (long long)B->n_cols, (long long)nb);
        return PFZ_ERR_UNSUPPORTED;
    }
    pfz_index *ix = new pfz_index();
    ix->ctx = ctx;
    ix->n_rows = B->n_rows;
    ix->n_cols = B->n_cols;
    ix->nnz = B->nnz;
    ix->block_cols = kC;
    ix->n_blocks = (int32_t)nb;
    PFZ_TRY(pool_alloc(ctx, &ix->tab, (size_t)(slots + 2) * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &ix->post, (size_t)(B->nnz > 0 ? B->nnz : 1) * sizeof(int2)));
    PFZ_HIP(hipMemsetAsync(ix->tab, 0, (size_t)(slots + 2) * sizeof(int32_t), ctx->stream));
    if (B->n_rows > 0 && B->nnz > 0) {
        const unsigned grid = (unsigned)((B->n_rows * 16 + 255) / 256);
        {
            ProfScope ps(ctx, "k_index_count");
            hipLaunchKernelGGL(k_index_count, dim3(grid), dim3(256), 0, ctx->stream, B->indptr, B->indices,
                               (int32_t)B->n_rows, (int32_t)nb, ix->tab + 1);
        }
        // counts sit at tab[1 + i]; exclusive scan of tab[1..] -> tab[1+i] = start(i)
        PFZ_TRY(exclusive_scan_i32(ctx, ix->tab + 1, slots));
        {
            ProfScope ps(ctx, "k_index_fill");
            // the fill advances tab[1+i] to end(i) = start(i+1); tab[0] = 0 = start(0)
            hipLaunchKernelGGL(k_index_fill, dim3(grid), dim3(256), 0, ctx->stream, B->indptr, B->indices, B->data,
                               (int32_t)B->n_rows, (int32_t)nb, ix->tab + 1, ix->post);
        }
        PFZ_HIP(hipGetLastError());
    }
    *out = ix;
    return PFZ_OK;
}

void pfz_index_free(pfz_index *ix)
{
    if (!ix) return;
    if (ix->ctx) (void)hipSetDevice(ix->ctx->device);
    if (ix->tab) pool_free(ix->tab);
    if (ix->post) pool_free(ix->post);
    delete ix;
}

int pfz_index_info(const pfz_index *ix, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int64_t *block_cols,
                   int64_t *n_blocks, int64_t *table_bytes)
{
    PFZ_REQUIRE(ix, "pfz_index_info: NULL index");
    if (n_rows) *n_rows = ix->n_rows;
    if (n_cols) *n_cols = ix->n_cols;
    if (nnz) *nnz = ix->nnz;
    if (block_cols) *block_cols = ix->block_cols;
    if (n_blocks) *n_blocks = ix->n_blocks;
    if (table_bytes) *table_bytes = (ix->n_cols * ix->n_blocks + 2) * (int64_t)sizeof(int32_t);
    return PFZ_OK;
}

int pfz_cossim_topn(pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int32_t ntop, float lower_bound,
                    int32_t exclude_diag, int64_t diag_offset, pfz_topn *out)
{
    PFZ_REQUIRE(ctx && ix && A && out, "pfz_cossim_topn: NULL argument");
    PFZ_REQUIRE(ntop >= 1, "pfz_cossim_topn: ntop must be >= 1 (got %d)", ntop);
    if (ntop > kMaxTop) {
        set_error("pfz_cossim_topn: ntop=%d exceeds the kernel's limit of %d", ntop, kMaxTop);
        return PFZ_ERR_UNSUPPORTED;
    }
    PFZ_REQUIRE(A->n_cols == ix->n_cols, "pfz_cossim_topn: from-matrix has %lld columns, index has %lld",
                (long long)A->n_cols, (long long)ix->n_cols);
    PFZ_REQUIRE(out->n_rows == A->n_rows && out->ntop == ntop, "pfz_cossim_topn: result buffer is %lldx%d, need %lldx%d",
                (long long)out->n_rows, out->ntop, (long long)A->n_rows, ntop);
    PFZ_REQUIRE(lower_bound == lower_bound, "pfz_cossim_topn: lower_bound is NaN");
    if (A->n_rows == 0) return PFZ_OK;
    PFZ_HIP(hipSetDevice(ctx->device));
    if (lower_bound < 0.f) lower_bound = 0.f;
    const int64_t max_grid = (int64_t)ctx->prop.multiProcessorCount * 16 * 64;
    const unsigned grid = (unsigned)(A->n_rows < max_grid ? A->n_rows : max_grid);
    {
        ProfScope ps(ctx, "k3_cossim_topn");
        hipLaunchKernelGGL(k3_cossim_topn_kernel, dim3(grid), dim3(64), 0, ctx->stream, A->indptr, A->indices, A->data,
                           (int32_t)A->n_rows, ix->tab, ix->post, ix->n_blocks, ntop, lower_bound, exclude_diag,
                           diag_offset, out->idx, out->val);
    }
    PFZ_HIP(hipGetLastError());
    return PFZ_OK;
}

int pfz_cossim_topn_host(pfz_ctx *ctx, int64_t n_from, int64_t n_to, int64_t n_cols, const int64_t *from_indptr,
                         const int32_t *from_indices, const float *from_data, const int64_t *to_indptr,
                         const int32_t *to_indices, const float *to_data, int32_t ntop, float lower_bound,
                         int32_t exclude_diag, int32_t *out_idx, float *out_val)
{
    PFZ_REQUIRE(ctx && out_idx && out_val, "pfz_cossim_topn_host: NULL argument");
    pfz_csr *A = nullptr, *B = nullptr;
    pfz_index *ix = nullptr;
    pfz_topn *res = nullptr;
    int rc = pfz_csr_upload(ctx, n_from, n_cols, from_indptr, from_indices, from_data, &A);
    if (rc == PFZ_OK) rc = pfz_csr_upload(ctx, n_to, n_cols, to_indptr, to_indices, to_data, &B);
    if (rc == PFZ_OK) rc = pfz_index_build(ctx, B, &ix);
    if (rc == PFZ_OK) rc = pfz_topn_alloc(ctx, n_from, ntop, &res);
    if (rc == PFZ_OK) rc = pfz_cossim_topn(ctx, ix, A, ntop, lower_bound, exclude_diag, 0, res);
    if (rc == PFZ_OK) rc = pfz_topn_download(ctx, res, out_idx, out_val);
    pfz_topn_free(res);
    pfz_index_free(ix);
    pfz_csr_free(B);
    pfz_csr_free(A);
    return rc;
}

}  // extern "C"
