// K3 -- sparse cosine top-n: C = A * B^T (CSR x CSR^T) fused with the strict
// lower bound, the self-match diagonal removal and the per-row top-n.
//
// Replaces sparse_dot_topn.awesome_cossim_topn as called at reference
// polyfuzz/models/_utils.py:82, plus _utils.py:84-91,128-146.
//
// Data layout in HBM
//   from side : CSR (indptr int32, indices int32 sorted, data fp32)
//   to side   : inverted index.  The to-rows are cut into blocks of C rows
//               (C = 1024 / 2048 / 4096); the postings of n-gram id k in block b
//               -- (byte offset of the to-row's accumulator, fp32 value) pairs --
//               form list (k,b).  Every list is PADDED to a multiple of 16
//               postings with zero-valued entries, so the index is an array of
//               16-posting PIECES, each exactly one aligned 128-byte line:
//               piece p is post[16p .. 16p+16), list (k,b) is pieces
//               tab[k*nb+b] .. tab[k*nb+b+1).  tab is one int32 array of V*nb+1
//               piece offsets, so a from-row walks block b of all its n-grams by
//               streaming along one table row per n-gram.  Piece tab[V*nb] is an
//               all-zero dummy.
//
// Arithmetic: fixed point.  Every product a*b is computed in fp32, scaled by
// S = 2^30 / (norm bound) and truncated to int32; the accumulators are int32 and
// are updated with ds_add_u32.  Integer addition is associative, so the sum does
// not depend on the order in which postings arrive: results are bit-reproducible,
// exact ties (duplicate to-strings) stay exact ties, and the kernel is free to
// process postings in whatever order fills the lanes best -- and a padding entry
// (value 0) adds nothing.  |sum/S - exact| is below 13 * 2^-30 + fp32 product
// rounding (~3e-8), far inside the 1e-5 budget.
// (Measured on MI355X, tools/ubench/lds_atomic.hip: ds_add_u32 6.6 lanes/clk/CU,
// plain LDS read+fadd+write 3.6, ds_add_f32 0.31 -- the float atomic is unusable.)
//
// Kernel (one one-wave workgroup == one from-row at a time; the hardware
// dispatcher load-balances the very skewed rows)
//   for each to-block b:
//     scatter: lane l owns n-gram l of the row and its (k,b) list; the lists'
//       pieces are numbered across all lists (DPP prefix sum) and dealt to the
//       four 16-lane quarters of the wave, 64 pieces per round: a round finds the
//       owner list of each of its 64 pieces once (LDS markers + DPP max-scan,
//       three ds_bpermute), then every step is v_add_dpp / global_load_dwordx2 /
//       v_mul_dpp / v_cvt / ds_add_u32 with the quarter's piece descriptor
//       broadcast by DPP row_newbcast -- no scalar work, no exec masking, one
//       full 128-byte line per quarter (see scatter_pieces).
//     sweep: lanes read acc as int4 pairs, write zeros back, and test the max of
//       8 sums against the running threshold (v_max3_i32); survivors go to a
//       candidate buffer as 64-bit keys sum<<32 | ~col, so an unsigned max is
//       "score desc, col asc".  When the buffer fills the wave keeps its ntop best
//       (rounds of wave-max) and raises the threshold to the ntop-th sum.
//   The final compaction leaves the sorted top-n; lanes write (idx, sum/S).
//
// Roofline: bound by LDS (atomics + sweep) and instruction issue; the index
// (tens of MB) is served by L2 / Infinity Cache, not HBM.
#include "k3_core.h"

#include <algorithm>
#include <math.h>
#include <stdlib.h>

namespace pfz {

constexpr int kMergeCap = 256;   // candidate keys per wave in k3_merge_slices
constexpr int kMaxTop = 1024;     // (beyond 128: a 1152-key candidate buffer -- half the resident workgroups -- and no to-slicing)

// ---------------------------------------------------------------------------
// inverted-index build (`block` to-rows per block)
//   cnt[k*nb + b]  = postings of list (k,b)                 (count kernels)
//   tab[k*nb + b]  = pieces of the list -> exclusive scan -> first piece
//   pad, then fill: post[16 * tab[k*nb+b] + position inside the list]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_index_count(const int32_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ indices, int32_t n_rows,
                                                      int32_t nb, int32_t block, int32_t *__restrict__ cnt)
{
    // one 16-lane group per to-row: rows have ~13 entries
    const int gid = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    if (gid >= n_rows) return;
    const int p0 = indptr[gid], p1 = indptr[gid + 1];
    const int b = gid / block;
    for (int p = p0 + sub; p < p1; p += 16) atomicAdd(&cnt[(int64_t)indices[p] * nb + b], 1);
}

// (global-atomics path of huge vocabularies) position inside the list = what is left of the count
__global__ __launch_bounds__(256) void k_index_fill(const int32_t *__restrict__ indptr,
                                                     const int32_t *__restrict__ indices,
                                                     const float *__restrict__ data, int32_t n_rows, int32_t nb,
                                                     int32_t block, int32_t *__restrict__ cnt,
                                                     const int32_t *__restrict__ tab, int2 *__restrict__ post)
{
    const int gid = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    if (gid >= n_rows) return;
    const int p0 = indptr[gid], p1 = indptr[gid + 1];
    const int b = gid / block;
    const int local = gid - b * block;
    for (int p = p0 + sub; p < p1; p += 16) {
        // the order inside one (k,b) list is irrelevant to the results (integer sums)
        const int64_t slot = (int64_t)indices[p] * nb + b;
        const int pos = atomicSub(&cnt[slot], 1) - 1;
        post[(int64_t)tab[slot] * kPiece + pos] = make_int2(local * 4, __float_as_int(data[p]));   // .x = byte offset into acc
    }
}

// pieces per list (before the scan).  sub != NULL (LDS-histogram path): the counts arrive per (list, sub-block) --
// kSub workgroups share a to-block -- and are turned in place into each sub-block's first position inside the list.
constexpr int kSub = 4;
// heavy != NULL: the lists of at least bank_min postings are noted for k_index_bank_order (heavy[0] = how many, then the slots)
__global__ __launch_bounds__(256) void k_index_pieces(int32_t *__restrict__ cnt, int32_t *__restrict__ sub, int64_t slots,
                                                       int32_t *__restrict__ tab, int32_t *__restrict__ heavy, int32_t heavy_cap,
                                                       int32_t bank_min)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int c = 0;
    if (i == 0) tab[-1] = 1;           // piece 0, in front of every list: the all-zero dummy (the scan runs from there)
    if (i < slots) {
        if (sub) {
            int4 v = ((const int4 *)sub)[i];
            c = v.x + v.y + v.z + v.w;
            if (c) ((int4 *)sub)[i] = make_int4(0, v.x, v.x + v.y, v.x + v.y + v.z);
            cnt[i] = c;
        } else {
            c = cnt[i];
        }
        tab[i] = (c + kPiece - 1) / kPiece;
    } else if (i < slots + 2) {
        tab[i] = 0;                        // (the table's tail: no memset in front of this kernel -- the grid covers slots + kPiece threads)
    }
    // one device-scope atomic per wave that has heavy lists (one per list made this kernel 5.5 -> 15.5 us at 650k slots)
    const bool h = heavy && i < slots && c >= bank_min;
    const uint64_t m = __ballot(h);
    if (m) {
        const int lane = threadIdx.x & 63, first = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == first) base = atomicAdd(&heavy[0], __popcll(m));
        base = __shfl(base, first, 64);
        const int at = base + __popcll(m & ((1ull << lane) - 1ull));
        if (h && at < heavy_cap) heavy[1 + at] = (int32_t)i;
    }
}

// Padding entries of every list (positions count .. 16*pieces) and the dummy piece (piece 0): value 0
// -- they add nothing -- at accumulator offsets spread over 64 cells, so the padding lanes of one
// ds_add_u32 do not pile up on one LDS address.
// pblk != NULL: the to-block of every piece of the list is noted as well (slot i = n-gram * nb + block)
__global__ __launch_bounds__(256) void k_index_pad(const int32_t *__restrict__ cnt, const int32_t *__restrict__ tab,
                                                    int64_t slots, int2 *__restrict__ post,
                                                    uint16_t *__restrict__ pblk, int32_t nb, const int32_t *__restrict__ nnz_dev,
                                                    int32_t *__restrict__ nnz_host)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) __hip_atomic_store(nnz_host, *nnz_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (the postings' count, for whoever asks the host)
    if (i < kPiece) post[i] = make_int2((int)i * 4, 0);      // piece 0: the dummy
    if (i == 0 && pblk) pblk[0] = 0;
    if (i >= slots) return;
    const int c = cnt[i];
    if (c == 0) return;
    if (pblk) {
        const uint16_t b = (uint16_t)((uint32_t)i % (uint32_t)nb);      // (slots < 2^31, pfz_index_build)
        const int first = tab[i], last = first + (c + kPiece - 1) / kPiece;
        for (int p = first; p < last; ++p) pblk[p] = b;
    }
    if ((c & (kPiece - 1)) == 0) return;
    const int64_t base = (int64_t)tab[i] * kPiece;
    const int end = (c + kPiece - 1) & ~(kPiece - 1);
    for (int p = c; p < end; ++p) post[base + p] = make_int2(((((int)i & 3) << 4) | (p & 15)) * 4, 0);
}

// The count / fill passes WITHOUT global atomics (device-scope atomics cost ~85 ns each in aggregate on
// this part -- 1.3 M of them were 90 % of the index build): one workgroup per to-block keeps a
// histogram of the block's n-grams in LDS, two 16-bit counters per word (a block has at most 8192
// rows), for vocabularies of up to 2 * kHistWords n-grams.
//   count: cnt[k*nb + b] = number of rows of block b that contain k      (cnt zeroed beforehand)
//   fill : position inside the (k,b) list = value the LDS counter had before this row's increment
__global__ __launch_bounds__(1024) void k_index_count_lds(const int32_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices, int32_t n_rows,
                                                           int32_t nb, int32_t block, int32_t words,
                                                           int32_t *__restrict__ sub_cnt /* [slots][kSub] */)
{
    __shared__ uint32_t h[kHistWords];
    for (int t = threadIdx.x; t < words; t += 1024) h[t] = 0u;
    __syncthreads();
    // workgroup = one quarter of a to-block's rows: the scattered 8-byte stores of the fill pass are what limits
    // these two kernels (a workgroup per block kept 49 of 256 CUs busy at 100k rows), so a block is shared by kSub
    const int b = blockIdx.x / kSub, q = blockIdx.x % kSub;
    const int per = block / kSub;
    const int lo = b * block + q * per, hi = min(n_rows, lo + per);
    // a 16-lane group takes 16 consecutive to-rows at a time: row bounds fetched in parallel by the lanes,
    // then the first 16 entries of all 16 rows loaded back to back (no dependent round trip per row)
    const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int lane0 = (threadIdx.x & 63) & ~15;
    for (int rb = lo + grp * 16; rb < hi; rb += 64 * 16) {
        const int mine = rb + sub;
        int q0 = 0, q1 = 0;
        if (mine < hi) {
            q0 = indptr[mine];
            q1 = indptr[mine + 1];
        }
        int kk[16], p0i[16], p1i[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            p0i[i] = __shfl(q0, lane0 + i, 64);
            p1i[i] = __shfl(q1, lane0 + i, 64);
            kk[i] = p0i[i] + sub < p1i[i] ? indices[p0i[i] + sub] : 0;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (p0i[i] + sub < p1i[i]) atomicAdd(&h[kk[i] >> 1], 1u << ((kk[i] & 1) * 16));
        for (int i = 0; i < 16; ++i) {            // rows with more than 16 entries
            const int a = __shfl(q0, lane0 + i, 64), e = __shfl(q1, lane0 + i, 64);
            for (int p = a + 16 + sub; p < e; p += 16) {
                const int k = indices[p];
                atomicAdd(&h[k >> 1], 1u << ((k & 1) * 16));
            }
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < words; w += 1024) {
        const uint32_t v = h[w];
        if (v & 0xffffu) sub_cnt[((int64_t)(2 * w) * nb + b) * kSub + q] = (int32_t)(v & 0xffffu);
        if (v >> 16) sub_cnt[((int64_t)(2 * w + 1) * nb + b) * kSub + q] = (int32_t)(v >> 16);
    }
}

__global__ __launch_bounds__(1024) void k_index_fill_lds(const int32_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ indices,
                                                          const float *__restrict__ data, int32_t n_rows, int32_t nb,
                                                          int32_t block, int32_t words,
                                                          const int32_t *__restrict__ tab /* first pieces */,
                                                          const int32_t *__restrict__ sub_first /* [slots][kSub] */,
                                                          int2 *__restrict__ post)
{
    __shared__ uint32_t h[kHistWords];
    for (int t = threadIdx.x; t < words; t += 1024) h[t] = 0u;
    __syncthreads();
    const int b = blockIdx.x / kSub, q = blockIdx.x % kSub;
    const int per = block / kSub;
    const int blk0 = b * block;
    const int lo = blk0 + q * per, hi = min(n_rows, lo + per);
    const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int lane0 = (threadIdx.x & 63) & ~15;
    // one posting: its list position = first position of this sub-block in the list + the value the LDS counter had
    // before this row (the order inside one (k,b) list is irrelevant to the results: integer sums)
    auto place = [&](int k, float v, int row, int64_t start) {
        const int sh = (k & 1) * 16;
        const uint32_t old = atomicAdd(&h[k >> 1], 1u << sh);
        post[start + (int)((old >> sh) & 0xffffu)] = make_int2((row - blk0) * 4, __float_as_int(v));   // .x = byte offset into acc
    };
    auto start_of = [&](int k) -> int64_t {
        const int64_t slot = (int64_t)k * nb + b;
        return (int64_t)tab[slot] * kPiece + sub_first[slot * kSub + q];
    };
    for (int rb = lo + grp * 16; rb < hi; rb += 64 * 16) {   // 16 consecutive rows per group, as in the count kernel
        const int mine = rb + sub;
        int q0 = 0, q1 = 0;
        if (mine < hi) {
            q0 = indptr[mine];
            q1 = indptr[mine + 1];
        }
        int kk[16];
        int64_t st[16];
        float vv[16];
        bool ok[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int a = __shfl(q0, lane0 + i, 64), e = __shfl(q1, lane0 + i, 64);
            ok[i] = a + sub < e;
            kk[i] = ok[i] ? indices[a + sub] : 0;
            vv[i] = ok[i] ? data[a + sub] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = ok[i] ? start_of(kk[i]) : 0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (ok[i]) place(kk[i], vv[i], rb + i, st[i]);
        for (int i = 0; i < 16; ++i) {            // rows with more than 16 entries
            const int a = __shfl(q0, lane0 + i, 64), e = __shfl(q1, lane0 + i, 64);
            for (int p = a + 16 + sub; p < e; p += 16) {
                const int k = indices[p];
                place(k, data[p], rb + i, start_of(k));
            }
        }
    }
}

// C to-rows per block; the (one-wave) workgroup owns one from-row at a time.
// kBounded (the deep top-n, ntop > kMaxTop: pfz_cossim_topn_rows): only keys below ub[row] count -- "the next ntop after the
// last key of the pass before" --, and the row's last key (0: the row is exhausted) replaces ub[row] at the end
template <int C, int kCap, bool kBounded = false>
__global__ __launch_bounds__(64) void k3_cossim_topn_kernel(
    const int32_t *__restrict__ a_indptr, const int32_t *__restrict__ a_idx, const float *__restrict__ a_val,
    int32_t n_a, const int32_t *__restrict__ tab, const int2 *__restrict__ post, int32_t nb, int32_t n_pieces,
    int32_t ntop, int32_t thr0, float scale, float inv_scale, int32_t exclude_diag, int64_t diag_offset,
    int32_t *__restrict__ out_idx, float *__restrict__ out_val, int32_t ablate, int32_t n_slices,
    uint64_t *__restrict__ part_keys, uint64_t *__restrict__ ub = nullptr)
{
    // accumulators first: this struct is the kernel's only LDS object, so they land at LDS address 0 and a
    // posting's byte offset IS its LDS address (run_steps)
    __shared__ __attribute__((aligned(16))) struct {
        int acc[C];
        uint64_t cand[kCap];
    } sm;
    static_assert(kCap >= 64 + 32, "the last 32 keys of the candidate buffer double as the scatter's marker scratch");
    int *const acc = sm.acc;
    uint64_t *const cand = sm.cand;
    // push4() compacts as soon as more than kCap - 64 keys are buffered, so while a block is scattered the
    // last 64 keys (512 B) are free: 64 marker ints live in the last 256 B
    int *const mark = (int *)(sm.cand + kCap) - 64;
    if ((uint32_t)(uintptr_t)sm.acc != 0u) __builtin_trap();   // layout assumption of run_steps()
    const int lane = threadIdx.x;
    int4 *acc4 = (int4 *)acc;
    constexpr int N4 = C / 4;   // int4 slots swept by the wave
    // a zero register for the whole kernel: as an asm result the compiler cannot re-materialise it
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    for (int t = lane; t < C / 4; t += 64) acc4[t] = make_int4(0, 0, 0, 0);
    wave_sync();
    const char *post_bytes = (const char *)post;
    const int src4 = (4 * (lane & 15) + (lane >> 4)) * 4;   // bpermute index of the window position this lane processes
    const int sub8 = (lane & 15) * 8;                       // byte offset of this lane's posting inside its piece
    const int dummy_addr = 0;                               // the all-zero piece is piece 0
    (void)n_pieces;

    // Work item = (from-row, to-slice).  The to-blocks are cut into n_slices contiguous
    // ranges and item i works on slice i % n_slices (small query batches: fill the chip).
    const int per_slice = (nb + n_slices - 1) / n_slices;
    for (int item = blockIdx.x; item < n_a * n_slices; item += gridDim.x) {
        const int row = item / n_slices, slice = item - row * n_slices;
        const int b_lo = slice * per_slice, b_hi = min(nb, b_lo + per_slice);
        const int p0 = a_indptr[row], p1 = a_indptr[row + 1];
        const int nnz = p1 - p0;
        const int64_t self_col64 = (int64_t)row + diag_offset;
        const int self_col = (exclude_diag && self_col64 >= 0 && self_col64 < 0x7fffffff) ? (int)self_col64 : -1;
        TopState st;
        st.cnt = 0;
        st.thr = thr0;

        // Block order: a self-match starts with the block that holds the from-row itself and wraps around.  On a
        // sorted list (the reference's company names are) a row's best matches sit next to it, so the threshold is
        // high from the first block on and far fewer sums are parked as candidates in the others; any order gives the
        // same result (the final selection is by key).  PFZ_K3_EXP 8: always start at the first block.
        const int n_blk = b_hi - b_lo;
        int b_first = b_lo;
#if PFZ_K3_EXP != 8
        if (self_col >= 0 && self_col / C >= b_lo && self_col / C < b_hi) b_first = self_col / C;
#endif
        // registers for the first 64 n-grams of the row (covers almost every row);
        // the two offset-table entries of the next block are always one block ahead in flight
        int cur0 = 0, nxt0 = 0;
        float as0 = 0.f;
        const bool have0 = lane < nnz;
        const int32_t *trow = tab;
        if (have0) {
            as0 = a_val[p0 + lane] * scale;
            trow = tab + (int64_t)a_idx[p0 + lane] * nb;
            cur0 = trow[b_first];
            nxt0 = trow[b_first + 1];
        }

        bool warmed = ablate == 3 || kBounded;   // (3: timing experiment without the threshold warm start; the warm start counts sums a bound may exclude)
        const uint64_t ub_row = kBounded ? ub[row] : ~0ull;
        for (int it = 0, b = b_first; it < n_blk; ++it) {
            const int s = cur0, e = have0 ? nxt0 : cur0;
            const int b_next = b + 1 < b_hi ? b + 1 : b_lo;
            bool touched = __ballot(e > s) != 0;
            if (touched && ablate != 1) scatter_pieces(acc, post_bytes, mark, e - s, s, as0, lane, src4, sub8, dummy_addr);
            // prefetch of the next block's table entries (tab has V*nb+2 slots): issued AFTER the block's posting
            // loads -- the compiler drains vmcnt before a round's loads -- and covered by the sweep below
            if (have0 && it + 1 < n_blk) {
                cur0 = trow[b_next];
                nxt0 = trow[b_next + 1];
            }
            for (int c0 = p0 + 64; c0 < p1; c0 += 64) {  // rows with more than 64 n-grams
                int s2 = 0, e2 = 0;
                float as2 = 0.f;
                if (c0 + lane < p1) {
                    const int k = a_idx[c0 + lane];
                    as2 = a_val[c0 + lane] * scale;
                    s2 = tab[(int64_t)k * nb + b];
                    e2 = tab[(int64_t)k * nb + b + 1];
                }
                if (__ballot(e2 > s2)) {
                    touched = true;
                    scatter_pieces(acc, post_bytes, mark, e2 - s2, s2, as2, lane, src4, sub8, dummy_addr);
                }
            }
            if (touched && ablate != 2) {
                wave_sync();      // the wave's LDS operations execute in order: the block's updates are in acc
                if (!warmed) {
                    warmed = true;
                    if (ntop <= kWarmMaxTop) {
                        const int t = warm_threshold<N4>(acc4, 0, ntop + (self_col >= 0 ? 1 : 0), lane);
                        st.thr = t > st.thr ? t : st.thr;
                    }
                }
                sweep_block<N4, kCap, kBounded>(acc4, cand, st, b * C, self_col, ntop, lane, zero, ub_row);
                wave_sync();      // acc is zero again
            }
            b = b_next;
        }

        compact<kCap>(cand, st, ntop, lane);
        if (kBounded && lane == 0) ub[row] = st.cnt == ntop ? cand[ntop - 1] : 0ull;     // (fewer than ntop: nothing is left below)
        for (int r = lane; r < ntop; r += 64) {
            const uint64_t key = r < st.cnt ? cand[r] : 0ull;
            if (n_slices > 1) {   // partial result of this slice; k3_merge_slices finishes the row
                part_keys[((int64_t)row * n_slices + slice) * ntop + r] = key;
            } else {
                out_idx[(int64_t)row * ntop + r] = key ? (int32_t)(~(uint32_t)key) : -1;
                out_val[(int64_t)row * ntop + r] = key ? (float)(int32_t)(uint32_t)(key >> 32) * inv_scale : 0.f;
            }
        }
        wave_sync();    // cand is reused by the next item
    }
}

// Bank order (round 4).  A ds_add_u32 of the scatter covers 64 consecutive postings of a heavy list, and the LDS serialises
// lanes that hit the same bank (word address mod 32 = to-row mod 32): in arrival order a list's rows are a random subset
// of the block and 38 % of the kernel's LDS cycles are bank conflicts (rocprofv3: 489 M of 1292 M per 100k x 100k launch).
// The order of the postings inside a list is free (integer sums), so the lists with at least kBankMin postings are
// re-dealt: posting i gets the rank j it has among the postings of its bank, and the list is written out by (j, bank) --
// banks cycle 0, 1, 2 ... 31, 0, 1 ... as long as every bank has postings left, so any 64 consecutive postings hit every
// bank about twice, wherever a round's piece numbering happens to cut.  One wave per heavy list; results are unchanged.
constexpr int kBankMin = 64, kBankCap = 1024;     // (kBankMin: 16 ... 128 gave the same K3 time, round 4)
__global__ __launch_bounds__(256) void k_index_bank_order(const int32_t *__restrict__ cnt, const int32_t *__restrict__ tab,
                                                           const int32_t *__restrict__ heavy, int32_t heavy_cap,
                                                           int2 *__restrict__ post)
{
    __shared__ int2 s_buf[4][kBankCap];
    __shared__ int s_cnt[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int2 *buf = s_buf[wave];
    int *bc = s_cnt[wave];
    // one wave per heavy list (the lists of one heavy n-gram are neighbours in slot order, and the list k_index_pieces
    // left is in no particular order: taking them round-robin spreads them over the waves)
    const int n_heavy = min(heavy[0], heavy_cap);
    const int n_waves = (int)gridDim.x * 4;
    for (int h = (int)blockIdx.x * 4 + wave; h < n_heavy; h += n_waves) {
        {
            const int64_t slot = heavy[1 + h];
            const int c = min(cnt[slot], kBankCap);
            const int64_t base = (int64_t)tab[slot] * kPiece;
            if (lane < 32) bc[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // rank of every posting inside its bank (any order inside a bank will do)
            for (int i = lane; i < c; i += 64) {
                int2 pe = post[base + i];
                const int b = (pe.x >> 2) & 31;
                const int j = atomicAdd(&bc[b], 1);
                buf[i] = make_int2(pe.x, pe.y);
                // (the rank travels in the upper bits of the offset: offsets are < 4 * 4096 = 2^14)
                buf[i].x = pe.x | (j << 16);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int my_cnt = lane < 32 ? bc[lane] : 0;
            for (int i = lane; i < c + 63 - (c + 63) % 64; i += 64) {
                const bool on = i < c;
                int2 pe = on ? buf[i] : make_int2(0, 0);
                const int j = (int)((uint32_t)pe.x >> 16), b = (pe.x >> 2) & 31;
                // position by (j, bank): the postings of ranks below j, plus the banks below b that reach rank j
                int pos = 0;
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    const int cq = __shfl(my_cnt, q, 64);
                    pos += min(cq, j) + ((q < b && cq > j) ? 1 : 0);
                }
                if (on) post[base + pos] = make_int2(pe.x & 0xffff, pe.y);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Merge the n_slices partial top-n key lists of every from-row (one wave per row).
__global__ __launch_bounds__(256) void k3_merge_slices(const uint64_t *__restrict__ part_keys, int32_t n_a,
                                                       int32_t n_slices, int32_t ntop, float inv_scale,
                                                       int32_t *__restrict__ out_idx, float *__restrict__ out_val)
{
    constexpr int kCap = kMergeCap;
    __shared__ __attribute__((aligned(16))) uint64_t cand_all[4][kCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n_a) return;
    uint64_t *cand = cand_all[wave];
    TopState st;
    st.cnt = 0;
    st.thr = 0;
    // the row's n_slices x ntop keys lie back to back (zeros = no entry): 64 per load, whatever slice they belong to -- a loop
    // over the slices made a single query wait for 49 dependent loads (11 us of its 35 us of kernels)
    const uint64_t *src = part_keys + (int64_t)row * n_slices * ntop;
    const int total = n_slices * ntop;
    for (int e0 = 0; e0 < total; e0 += 64) {
        const uint64_t key = e0 + lane < total ? src[e0 + lane] : 0ull;
        const uint64_t mk = __ballot(key != 0ull);
        if (key) cand[st.cnt + __popcll(mk & ((1ull << lane) - 1ull))] = key;
        st.cnt += __popcll(mk);
        if (st.cnt > kCap - 64) compact<kCap>(cand, st, ntop, lane, false);
    }
    compact<kCap>(cand, st, ntop, lane);
    for (int r = lane; r < ntop; r += 64) {
        const uint64_t key = r < st.cnt ? cand[r] : 0ull;
        out_idx[(int64_t)row * ntop + r] = key ? (int32_t)(~(uint32_t)key) : -1;
        out_val[(int64_t)row * ntop + r] = key ? (float)(int32_t)(uint32_t)(key >> 32) * inv_scale : 0.f;
    }
}

// deep top-n: the columns [col0, col0 + w) of a result whose rows are ntop wide <- a w-wide pass result
__global__ __launch_bounds__(256) void k_topn_place(const int32_t *__restrict__ t_idx, const float *__restrict__ t_val, int64_t n_rows,
                                                    int32_t w, int32_t ntop, int32_t col0, int32_t *__restrict__ out_idx,
                                                    float *__restrict__ out_val)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows * w) return;
    const int64_t r = i / w;
    const int c = (int)(i - r * w);
    out_idx[r * ntop + col0 + c] = t_idx[i];
    out_val[r * ntop + col0 + c] = t_val[i];
}

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

int index_ready(const pfz_index *ix)
{
    int32_t v = 0;
    if (ix->pieces_lazy.pending) {
        PFZ_TRY(lazy_get(ix->ctx, &ix->pieces_lazy, &v));
        ix->n_pieces = v - 1;           // (the count includes the dummy)
    }
    if (ix->nnz_lazy.pending) {
        PFZ_TRY(lazy_get(ix->ctx, &ix->nnz_lazy, &v));
        ix->nnz = v;
    }
    return PFZ_OK;
}

}  // namespace pfz

using namespace pfz;

extern "C" {

int pfz_index_build(pfz_ctx *ctx, const pfz_csr *B, pfz_index **out)
{
    PFZ_REQUIRE(ctx && B && out, "pfz_index_build: NULL argument");
    PFZ_HIP(hipSetDevice(ctx->device));
    // to-rows per block (tuning knob PFZ_K3_BLOCK, tools/sweep_k3.sh).  2048 is the best point: 100k to-rows 2.94 ms against
    // 3.75 with 4096 (LDS occupancy).  For a to-side of a million rows the row-major kernel used to prefer 4096 (index
    // 210 MB instead of 267: 47 ms against 52 for the 125k x 1M shard, fabric-bound); the lock-step kernel (k3_lockstep.hip)
    // serves the postings from L2 and is fastest on 2048-row blocks (33 ms), and small query batches against the big
    // index are as fast or faster with 2048 (600 rows 0.39 ms against 0.44, 5 000 rows 2.39 against 2.52)
    int block = env_int("PFZ_K3_BLOCK", 2048);
    if (block != 1024 && block != 1536 && block != 2048 && block != 4096) block = 2048;
    const int64_t nb = (B->n_rows + block - 1) / block;
    const int64_t slots = B->n_cols * nb;
    if (slots >= ((int64_t)1 << 31) - 2) {
        set_error("pfz_index_build: vocabulary %lld x %lld to-blocks exceeds the int32 offset table",
                  (long long)B->n_cols, (long long)nb);
        return PFZ_ERR_UNSUPPORTED;
    }
    Owner<pfz_index, pfz_index_free> ix(new pfz_index());
    ix->ctx = ctx;
    ix->n_rows = B->n_rows;
    ix->n_cols = B->n_cols;
    ix->block_cols = block;
    ix->n_blocks = (int32_t)nb;
    ix->max_norm = B->max_norm;
    ix->src_serial = B->serial;
    struct Tmp {
        int32_t *p = nullptr;
        ~Tmp() { if (p) pool_free(p); }
    } cnt, sub;
    // (a device-vectorised matrix does not know its number of non-zeros on the host yet -- csr_nnz() would wait for it; the
    // build runs on whatever it is, zero included)
    const bool any = B->n_rows > 0;
    // per-block LDS histograms when the vocabulary fits (PFZ_NO_LDS_HIST=1 forces the global-atomics
    // path of huge vocabularies: tests)
    const bool lds_hist = B->n_cols <= 2 * (int64_t)kHistWords && !getenv("PFZ_NO_LDS_HIST");
    // tab_base[0] = 1 (the dummy piece), tab_base[1 + i] = pieces of list i; after the scan tab = tab_base + 1 holds every
    // list's first piece and tab[slots] the number of pieces, dummy included
    PFZ_TRY(pool_alloc(ctx, &ix->tab_base, (size_t)(slots + 3) * sizeof(int32_t)));
    ix->tab = ix->tab_base + 1;
    PFZ_TRY(pool_alloc(ctx, &cnt.p, (size_t)(slots + 2) * sizeof(int32_t)));
    // (no memset of the table: k_index_pieces writes every entry, the dummy's and the two of the tail included)
    if (lds_hist && any) {      // counts per (list, sub-block); cnt itself is written by k_index_pieces
        PFZ_TRY(pool_alloc(ctx, &sub.p, (size_t)(slots + 1) * kSub * sizeof(int32_t)));
        PFZ_HIP(hipMemsetAsync(sub.p, 0, (size_t)slots * kSub * sizeof(int32_t), ctx->stream));
    } else {
        PFZ_HIP(hipMemsetAsync(cnt.p, 0, (size_t)(slots + 2) * sizeof(int32_t), ctx->stream));
    }
    // the heavy lists, for k_index_bank_order (only where the counts survive the fill: the LDS-histogram path)
    // ... and where K3 is long enough to pay for the pass: 100k x 100k K3 2.80 -> 2.63 ms for 0.02 ms, 125k x 1M 33.7 -> 33.0 ms
    // for 0.05; at 10k x 10k (K3 0.07 ms) it only costs.  PFZ_K3_BANK_ORDER=1 forces it (tests)
    const char *bo_env = getenv("PFZ_K3_BANK_ORDER");
    const bool bank_order = any && lds_hist && !getenv("PFZ_K3_NO_BANK_ORDER") &&
                            ((bo_env && atoi(bo_env) > 0) || (!bo_env && B->n_rows >= 32768));
    const int32_t bank_min = kBankMin;
    const int32_t heavy_cap = (int32_t)std::min<int64_t>(slots, (int64_t)1 << 22);
    Tmp heavy;
    if (bank_order) {
        PFZ_TRY(pool_alloc(ctx, &heavy.p, (size_t)(heavy_cap + 1) * sizeof(int32_t)));
        PFZ_HIP(hipMemsetAsync(heavy.p, 0, sizeof(int32_t), ctx->stream));
    }
    const int32_t words = (int32_t)((B->n_cols + 1) / 2);
    const unsigned row_grid = (unsigned)((B->n_rows * 16 + 255) / 256);
    const unsigned slot_grid = (unsigned)((slots + kPiece + 255) / 256);
    {
        ProfScope ps(ctx, "k_index_count");
        if (any && lds_hist)
            hipLaunchKernelGGL(k_index_count_lds, dim3((unsigned)nb * kSub), dim3(1024), 0, ctx->stream, B->indptr, B->indices,
                               (int32_t)B->n_rows, (int32_t)nb, block, words, sub.p);
        else if (any)
            hipLaunchKernelGGL(k_index_count, dim3(row_grid), dim3(256), 0, ctx->stream, B->indptr, B->indices,
                               (int32_t)B->n_rows, (int32_t)nb, block, cnt.p);
        hipLaunchKernelGGL(k_index_pieces, dim3(slot_grid), dim3(256), 0, ctx->stream, cnt.p, lds_hist && any ? sub.p : nullptr, slots,
                           ix->tab, heavy.p, heavy_cap, bank_min);
    }
    PFZ_TRY(exclusive_scan_i32(ctx, ix->tab_base, slots + 1, &ix->pieces_lazy));   // tab[i] = first piece of list i, tab[slots] = pieces + the dummy
    // The postings are allocated by a BOUND of the number of pieces (a list of c postings has ceil(c / 16) pieces: at most nnz / 16
    // + one per non-empty list), not by the count itself, WHILE THE BOUND IS SMALL (<= 2^20 pieces = 128 MB: every job whose K3 is
    // short enough to notice): waiting for the count idles the device until the host has enqueued the fill (50 us of a 0.37-ms
    // step at 10k x 10k).  A larger bound is several times the real index (1M to-rows: 1 GB against 267 MB, twice that with the
    // symmetric copy) and its K3 takes milliseconds: there the count is waited for, checked against the 4 GiB the kernels address
    // with 32-bit byte offsets, and the postings get their exact size.
    const int64_t nnz_bound = B->nnz_lazy.pending ? B->nnz_cap : B->nnz;
    int64_t piece_cap = nnz_bound / kPiece + std::min<int64_t>(slots, nnz_bound) + 1;
    if (piece_cap > kPieceBoundMax) {
        int32_t total = 0;
        PFZ_TRY(lazy_get(ctx, &ix->pieces_lazy, &total));
        ix->n_pieces = total - 1;       // (lazy_get cleared `pending`: index_ready() will not assign it any more)
        if (total - 1 >= (1 << 25) - 1) {
            set_error("pfz_index_build: %d index pieces (%lld postings padded to 16 per list and block) exceed the 4 GiB "
                      "the kernel addresses with 32-bit byte offsets", total - 1, (long long)csr_nnz(B));
            return PFZ_ERR_UNSUPPORTED;
        }
        piece_cap = total;
    }
    ix->piece_cap = piece_cap;
    PFZ_TRY(lazy_acquire(ctx, &ix->nnz_lazy));      // (written by k_index_pad: the matrix' last row bound, wherever the host stands with B's own count)
    PFZ_TRY(pool_alloc(ctx, &ix->post, (size_t)piece_cap * kPiece * sizeof(int2)));
    if (block == 2048 && nb >= 2 && nb < 65536)      // (what the symmetric form of a self-match reads: k3_symmetric.hip)
        PFZ_TRY(pool_alloc(ctx, &ix->pblk, (size_t)piece_cap * sizeof(uint16_t)));
    {
        ProfScope ps(ctx, "k_index_fill");
        // (before the fill: the global-atomics fill counts cnt down)
        hipLaunchKernelGGL(k_index_pad, dim3(slot_grid), dim3(256), 0, ctx->stream, cnt.p, ix->tab, slots, ix->post, ix->pblk, (int32_t)nb,
                           B->indptr + B->n_rows, ix->nnz_lazy.slot);
        PFZ_TRY(lazy_mark(ctx, &ix->nnz_lazy));
        if (any && lds_hist)
            hipLaunchKernelGGL(k_index_fill_lds, dim3((unsigned)nb * kSub), dim3(1024), 0, ctx->stream, B->indptr, B->indices,
                               B->data, (int32_t)B->n_rows, (int32_t)nb, block, words, ix->tab, sub.p, ix->post);
        else if (any)
            hipLaunchKernelGGL(k_index_fill, dim3(row_grid), dim3(256), 0, ctx->stream, B->indptr, B->indices, B->data,
                               (int32_t)B->n_rows, (int32_t)nb, block, cnt.p, ix->tab, ix->post);
    }
    if (bank_order) {
        ProfScope ps(ctx, "k_index_bank_order");
        hipLaunchKernelGGL(k_index_bank_order, dim3((unsigned)ctx->prop.multiProcessorCount * 4), dim3(256), 0, ctx->stream, cnt.p,
                           ix->tab, heavy.p, heavy_cap, ix->post);
    }
    PFZ_HIP(hipGetLastError());
    *out = ix.release();
    return PFZ_OK;
}

void pfz_index_free(pfz_index *ix)
{
    if (!ix) return;
    if (ix->ctx) (void)hipSetDevice(ix->ctx->device);
    k3_sym_free(ix);
    for (LazyI32 *z : {&ix->pieces_lazy, &ix->nnz_lazy})
        if (z->pending || z->slot) {      // (the copy into the slot may still be in flight)
            if (z->ev) (void)hipEventSynchronize(z->ev);
            lazy_release(ix->ctx, z);
        }
    if (ix->tab_base) pool_free(ix->tab_base);
    if (ix->post) pool_free(ix->post);
    if (ix->pblk) pool_free(ix->pblk);
    delete ix;
}

int pfz_index_info(const pfz_index *ix, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int64_t *block_cols,
                   int64_t *n_blocks, int64_t *table_bytes)
{
    PFZ_REQUIRE(ix, "pfz_index_info: NULL index");
    PFZ_TRY(index_ready(ix));
    if (n_rows) *n_rows = ix->n_rows;
    if (n_cols) *n_cols = ix->n_cols;
    if (nnz) *nnz = ix->nnz;
    if (block_cols) *block_cols = ix->block_cols;
    if (n_blocks) *n_blocks = ix->n_blocks;
    if (table_bytes) *table_bytes = (ix->n_cols * ix->n_blocks + 2) * (int64_t)sizeof(int32_t);
    return PFZ_OK;
}

int pfz_index_pieces(const pfz_index *ix, int64_t *n_pieces, int64_t *piece_postings)
{
    PFZ_REQUIRE(ix, "pfz_index_pieces: NULL index");
    PFZ_TRY(index_ready(ix));
    if (n_pieces) *n_pieces = ix->n_pieces;
    if (piece_postings) *piece_postings = kPiece;
    return PFZ_OK;
}

int pfz_cossim_topn(pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int32_t ntop, float lower_bound,
                    int32_t exclude_diag, int64_t diag_offset, pfz_topn *out)
{
    PFZ_REQUIRE(A, "pfz_cossim_topn: NULL argument");
    return pfz_cossim_topn_rows(ctx, ix, A, 0, A->n_rows, ntop, lower_bound, exclude_diag, diag_offset, out);
}

int pfz_cossim_topn_rows(pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int64_t row_begin, int64_t row_end,
                         int32_t ntop, float lower_bound, int32_t exclude_diag, int64_t diag_offset, pfz_topn *out)
{
    PFZ_REQUIRE(ctx && ix && A && out, "pfz_cossim_topn: NULL argument");
    PFZ_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= A->n_rows,
                "pfz_cossim_topn_rows: rows [%lld, %lld) outside [0, %lld)", (long long)row_begin, (long long)row_end,
                (long long)A->n_rows);
    PFZ_REQUIRE(ntop >= 1, "pfz_cossim_topn: ntop must be >= 1 (got %d)", ntop);
    PFZ_REQUIRE(A->n_cols == ix->n_cols, "pfz_cossim_topn: from-matrix has %lld columns, index has %lld",
                (long long)A->n_cols, (long long)ix->n_cols);
    PFZ_REQUIRE(out->n_rows >= A->n_rows && out->ntop == ntop, "pfz_cossim_topn: result buffer is %lldx%d, need %lldx%d",
                (long long)out->n_rows, out->ntop, (long long)A->n_rows, ntop);
    PFZ_REQUIRE(lower_bound == lower_bound, "pfz_cossim_topn: lower_bound is NaN");
    const int64_t n_rows = row_end - row_begin;      // the from-rows of this launch
    if (n_rows == 0) return PFZ_OK;
    PFZ_HIP(hipSetDevice(ctx->device));
    float scale, inv_scale;
    int32_t thr0;
    k3_fixed_point(A, ix, lower_bound, &scale, &inv_scale, &thr0);
    if (ntop > kMaxTop) {
        // Deep top-n (the reference clips top_n to the number of distinct to-strings only, _utils.py:54-56): passes of kMaxTop.
        // A pass keeps, per row, the kMaxTop best keys BELOW the last key of the pass before (keys are distinct: sum << 32 |
        // ~column), so the passes continue each other exactly; a row that runs out of candidates is marked exhausted.
        struct Tmp {
            void *p = nullptr;
            ~Tmp() { if (p) pool_free(p); }
        } t_idx, t_val, ub;
        PFZ_TRY(pool_alloc(ctx, &t_idx.p, (size_t)n_rows * kMaxTop * sizeof(int32_t)));
        PFZ_TRY(pool_alloc(ctx, &t_val.p, (size_t)n_rows * kMaxTop * sizeof(float)));
        PFZ_TRY(pool_alloc(ctx, &ub.p, (size_t)n_rows * sizeof(uint64_t)));
        PFZ_HIP(hipMemsetAsync(ub.p, 0xff, (size_t)n_rows * sizeof(uint64_t), ctx->stream));
        const int64_t max_grid = (int64_t)ctx->prop.multiProcessorCount * 16 * 64 * 8;
        const unsigned grid = (unsigned)(n_rows < max_grid ? n_rows : max_grid);
        ProfScope ps(ctx, "k3_cossim_topn");
        for (int32_t col0 = 0; col0 < ntop; col0 += kMaxTop) {
            const int32_t w = ntop - col0 < kMaxTop ? ntop - col0 : kMaxTop;
#define PFZ_K3_DEEP(CC)                                                                                                   \
    hipLaunchKernelGGL((k3_cossim_topn_kernel<CC, 1152, true>), dim3(grid), dim3(64), 0, ctx->stream, A->indptr + row_begin,     \
                       A->indices, A->data, (int32_t)n_rows, ix->tab, ix->post, ix->n_blocks, 0, w, thr0, scale,      \
                       inv_scale, exclude_diag, diag_offset + row_begin, (int32_t *)t_idx.p, (float *)t_val.p, 0, 1,              \
                       (uint64_t *)nullptr, (uint64_t *)ub.p)
            switch (ix->block_cols) {
            case 1024: PFZ_K3_DEEP(1024); break;
            case 1536: PFZ_K3_DEEP(1536); break;
            case 2048: PFZ_K3_DEEP(2048); break;
            case 4096: PFZ_K3_DEEP(4096); break;
            default:
                set_error("pfz_cossim_topn: no kernel for block size %d", ix->block_cols);
                return PFZ_ERR_INVALID;
            }
#undef PFZ_K3_DEEP
            hipLaunchKernelGGL(k_topn_place, dim3((unsigned)((n_rows * w + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const int32_t *)t_idx.p, (const float *)t_val.p, n_rows, w, ntop, col0, out->idx + row_begin * ntop,
                               out->val + row_begin * ntop);
        }
        PFZ_HIP(hipGetLastError());
        return PFZ_OK;
    }
    // a list against itself: every unordered pair once (k3_symmetric.hip), whole jobs and ascending row ranges of one
    if (const int sym = k3_sym_wanted(ctx, ix, A, row_begin, row_end, ntop, thr0, scale, exclude_diag, diag_offset, out)) {
        ProfScope ps(ctx, "k3_cossim_topn");
        bool declined = false;
        const int rc = k3_sym_launch(ctx, ix, A, row_begin, row_end, ntop, thr0, scale, inv_scale, out, sym == 1, &declined);
        if (!declined) return rc;      // (declined: no memory for the session buffers -- the row-major kernel below needs none)
    }
    if (k3_lockstep_wanted(ctx, ix, n_rows, ntop)) {     // big to-sides: all waves on the same to-blocks (k3_lockstep.hip)
        ProfScope ps(ctx, "k3_cossim_topn");
        return k3_lockstep_launch(ctx, ix, A, row_begin, n_rows, ntop, thr0, scale, inv_scale, exclude_diag, diag_offset, out);
    }
    // to-side slices (tuning knob PFZ_K3_SLICES).  Slicing LOSES on a full job -- every slice restarts the
    // top-n threshold and pays the row set-up again -- and stays off there
    int n_slices = env_int("PFZ_K3_SLICES", 0);
    if (n_slices <= 0) {
        // auto: a small query batch (fit once / transform many, reference polyfuzz.py:234-240) cannot fill
        // 256 CUs x 18 resident workgroups with one workgroup per row -- cut the to-side until ~24 work items
        // per CU exist (measured: 1000 x 300k rows 1.07 -> 0.24 ms, 600 x 1M 2.9 -> 0.5 ms, 3000 x 100k
        // 0.36 -> 0.22 ms; a single query 1.0 ms; the 100k-row benchmark is unaffected)
        const int64_t want = (int64_t)ctx->prop.multiProcessorCount * 24;
        n_slices = n_rows >= want ? 1 : (int)((want + n_rows - 1) / n_rows);
    }
    if (ntop > 128) n_slices = 1;      // (k3_merge_slices holds kMergeCap = 256 keys)
    n_slices = n_slices < 1 ? 1 : (n_slices > ix->n_blocks ? (ix->n_blocks > 0 ? ix->n_blocks : 1) : n_slices);
    if (ix->n_blocks > 0) {
        // no empty trailing slice (nb = 9, 8 slices -> 2 blocks per slice -> only 5 slices have blocks): an empty
        // slice would still read its offset-table entries, past the end of the table for the last n-gram
        const int per_slice = (ix->n_blocks + n_slices - 1) / n_slices;
        n_slices = (ix->n_blocks + per_slice - 1) / per_slice;
    }
    uint64_t *part = nullptr;
    if (n_slices > 1) {
        PFZ_TRY(ensure_scratch(ctx, (size_t)n_rows * (size_t)n_slices * (size_t)ntop * sizeof(uint64_t)));
        part = (uint64_t *)ctx->scratch;
    }
    const int64_t items = n_rows * n_slices;
    const int64_t max_grid = (int64_t)ctx->prop.multiProcessorCount * 16 * 64 * 8;
    const unsigned grid = (unsigned)(items < max_grid ? items : max_grid / n_slices * n_slices);
    // candidate keys: room for ntop kept keys + the 64 one sweep step can add.  LDS per workgroup decides how
    // many from-rows a CU works on at once: 8 KiB of accumulators + 96 keys is 8960 B = 18 workgroups per CU
    const int cap = ntop <= 32 ? 96 : (ntop <= 64 ? 128 : (ntop <= 128 ? 256 : 1152));
#ifdef PFZ_EXPERIMENTS
    const int ablate = env_int("PFZ_K3_ABLATE", 0);   // timing experiments (variant builds only: tools/build_variant.sh -DPFZ_EXPERIMENTS): 1 = no scatter, 2 = no sweep, 3 = no warm start
#else
    const int ablate = 0;                              // (the shipped library has no knob that makes results wrong: tests/test_abi_cpu.py)
#endif
    {
        ProfScope ps(ctx, "k3_cossim_topn");
#define PFZ_K3_LAUNCH(CC, CAP)                                                                                  \
    hipLaunchKernelGGL((k3_cossim_topn_kernel<CC, CAP>), dim3(grid), dim3(64), 0, ctx->stream,                  \
                       A->indptr + row_begin, A->indices, A->data, (int32_t)n_rows, ix->tab, ix->post,          \
                       ix->n_blocks, 0, ntop, thr0, scale, inv_scale, exclude_diag,                  \
                       diag_offset + row_begin, out->idx + row_begin * ntop, out->val + row_begin * ntop,       \
                       ablate, n_slices, part)
#define PFZ_K3_CASE(CC)                              \
    case CC:                                         \
        if (cap == 96) PFZ_K3_LAUNCH(CC, 96);        \
        else if (cap == 128) PFZ_K3_LAUNCH(CC, 128); \
        else if (cap == 256) PFZ_K3_LAUNCH(CC, 256); \
        else PFZ_K3_LAUNCH(CC, 1152);                \
        break;
        switch (ix->block_cols) {
            PFZ_K3_CASE(1024)
            PFZ_K3_CASE(1536)
            PFZ_K3_CASE(2048)
            PFZ_K3_CASE(4096)
        default:
            set_error("pfz_cossim_topn: no kernel for block size %d", ix->block_cols);
            return PFZ_ERR_INVALID;
        }
#undef PFZ_K3_CASE
#undef PFZ_K3_LAUNCH
    }
    if (n_slices > 1) {
        ProfScope ps(ctx, "k3_merge_slices");
        hipLaunchKernelGGL(k3_merge_slices, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, ctx->stream, part,
                           (int32_t)n_rows, n_slices, ntop, inv_scale, out->idx + row_begin * ntop, out->val + row_begin * ntop);
    }
    PFZ_HIP(hipGetLastError());
    return PFZ_OK;
}

int pfz_cossim_topn_ranges(pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int32_t ntop, float lower_bound, int32_t exclude_diag,
                           int32_t n_ranges, const int64_t *range_ends, int32_t first_event, pfz_topn *out, const int32_t **host_idx,
                           const float **host_val)
{
    if (host_idx) *host_idx = nullptr;
    if (host_val) *host_val = nullptr;
    PFZ_REQUIRE(ctx && ix && A && out && range_ends, "pfz_cossim_topn_ranges: NULL argument");
    PFZ_REQUIRE(n_ranges >= 1 && first_event >= 0 && first_event + n_ranges <= kEventSlots,
                "pfz_cossim_topn_ranges: %d ranges from event slot %d do not fit the %d slots", n_ranges, first_event, kEventSlots);
    int64_t prev = 0;
    bool on_blocks = true;
    for (int32_t i = 0; i < n_ranges; ++i) {
        PFZ_REQUIRE(range_ends[i] > prev && range_ends[i] <= A->n_rows, "pfz_cossim_topn_ranges: range ends must ascend inside (0, %lld]",
                    (long long)A->n_rows);
        on_blocks = on_blocks && (i == n_ranges - 1 || range_ends[i] % 2048 == 0);
        prev = range_ends[i];
    }
    PFZ_REQUIRE(prev == A->n_rows, "pfz_cossim_topn_ranges: the last range must end at the matrix' last row");
    PFZ_REQUIRE(ntop >= 1 && lower_bound == lower_bound && A->n_cols == ix->n_cols && out->n_rows >= A->n_rows && out->ntop == ntop,
                "pfz_cossim_topn_ranges: arguments as for pfz_cossim_topn");
    PFZ_HIP(hipSetDevice(ctx->device));
    // a list against itself in the symmetric form: ONE pass-1 launch, the ranges handed on as they finish (k3_symmetric.hip)
    if (on_blocks && ntop <= kMaxTop && !getenv("PFZ_K3_NO_STREAMED")) {
        float scale, inv_scale;
        int32_t thr0;
        k3_fixed_point(A, ix, lower_bound, &scale, &inv_scale, &thr0);
        if (k3_sym_wanted(ctx, ix, A, 0, A->n_rows, ntop, thr0, scale, exclude_diag, 0, out) == 1) {
            // the mirror of the result in pinned host memory, if the caller wants one (kept with the context, grown on demand)
            int32_t *h_idx = nullptr;
            float *h_val = nullptr;
            if (host_idx && host_val) {
                const size_t cells = (size_t)A->n_rows * (size_t)ntop;
                const size_t bytes = cells * (sizeof(int32_t) + sizeof(float));
                if (bytes > ctx->mirror_bytes) {
                    PFZ_HIP(hipStreamSynchronize(ctx->stream));
                    if (ctx->stream3) PFZ_HIP(hipStreamSynchronize(ctx->stream3));
                    for (int q = 0; q < 3; ++q)
                        if (ctx->stream3x[q]) PFZ_HIP(hipStreamSynchronize(ctx->stream3x[q]));
                    if (ctx->mirror) PFZ_HIP(hipHostFree(ctx->mirror));
                    ctx->mirror = nullptr;
                    ctx->mirror_bytes = 0;
                    PFZ_HIP(hipHostMalloc((void **)&ctx->mirror, bytes + bytes / 8, hipHostMallocDefault));
                    ctx->mirror_bytes = bytes + bytes / 8;
                }
                h_idx = (int32_t *)ctx->mirror;
                h_val = (float *)(ctx->mirror + cells * sizeof(int32_t));
            }
            ProfScope ps(ctx, "k3_cossim_topn");
            bool declined = false;
            const int rc = k3_sym_launch_streamed(ctx, ix, A, ntop, thr0, scale, inv_scale, out, n_ranges, range_ends, first_event, h_idx, h_val,
                                                  &declined);
            if (!declined) {
                if (rc == PFZ_OK && h_idx) {
                    *host_idx = h_idx;
                    *host_val = h_val;
                }
                return rc;
            }
        }
    }
    // everything else: a launch per range, its event on the context's stream
    prev = 0;
    for (int32_t i = 0; i < n_ranges; ++i) {
        PFZ_TRY(pfz_cossim_topn_rows(ctx, ix, A, prev, range_ends[i], ntop, lower_bound, exclude_diag, 0, out));
        PFZ_HIP(hipEventRecord(ctx->events[first_event + i], ctx->stream));
        ctx->evt_want[first_event + i] = 0;
        prev = range_ends[i];
    }
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
