// K3, symmetric form -- the self-match of a list against itself (reference polyfuzz/models/_tfidf.py:109-116 with
// to_list=None -> _utils.py:82-91: awesome_cossim_topn(A, A.T, top_n + 1, min_sim), diagonal removed, per-row top-n),
// bit-identical to k3_cossim_topn.hip, with every UNORDERED pair of rows scored once.
//
// Why.  C = A * A^T is symmetric, and the fixed-point arithmetic of K3 keeps it symmetric bit for bit: the term of
// n-gram k in s(i,j) is trunc((a_ik * S) * a_jk), S a power of two, so (a_ik * S) * a_jk and (a_jk * S) * a_ik are the
// same fp32 product, and integer sums do not depend on their order.  The row-major kernel walks all nb to-blocks for
// every from-row: every pair is scattered and swept twice.  Here row j walks only the blocks from its own upwards;
// what it finds for itself it keeps as before, and what it finds for a row i of a higher block -- s(j,i) = s(i,j) --
// it hands to row i.  Scatter and sweep work halve; the price is a second filter in the sweep (is this sum a candidate
// for the ROW OF THE CELL?) and a small exchange through HBM.
//
// How.  Three passes of one kernel template (k3_sym_kernel<C, MODE>) + a merge:
//   0  every row x its OWN block (both directions of a pair inside a block are computed: no exchange there) -> the
//      row's first top-n and threshold, written to HBM (keys[row][ntop], thrv[row]) and, as the upper 16 bits of the
//      threshold, into thr16 in the order the sweep reads a block's cells.  On a sorted list (the reference's company
//      names are) a row's best matches sit next to it: the thresholds are high from the start.
//   1  row j x the blocks ABOVE its own: state restored, scatter as ever; the sweep tests every sum against the row's
//      own threshold (as ever) and, packed two to a word, the upper halves of the eight sums of a lane against the
//      upper halves of their eight rows' thresholds (v_perm_b32 x 4, v_pk_sub_i16 x 4, two v_bitop3, one compare).  The
//      16 bytes of thresholds per lane and step are loaded before the block's scatter (steps 0 / 1) and while those are
//      swept (steps 2 / 3): they are long there when the sweep wants them.
//      A hit (conservative: upper halves only) goes into a 128-entry LDS buffer as (sum, row of the cell); the buffer
//      is flushed -- one returning atomic per entry on push_cnt[i], one store into push_buf[i] -- when it is half full
//      and at the end of the row: about once per row (23 candidates per row on the 100 000 company names).  A row that
//      raises its own threshold publishes it (thrv, thr16) -- for whoever sees it: the stores reach the other XCDs' L2s
//      when the kernel ends, measured effect none; stale reads only let more candidates through.
//   merge  row i's own keys + what was pushed to it -> the sorted top-n (compact<> of k3_core.h, one wave per row).
//      A row that was pushed more than kSymPush candidates (195 of the 100 000 names: strings with hundreds of
//      near-equals elsewhere in the list; or a row without ntop positive matches in its own block, which keeps
//      threshold 0 and is sent every non-zero sum) is noted and
//   2  recomputed in full, the row-major way, by the same kernel -- in slices of the to-blocks, whose partial lists a
//      second merge joins: a whole row is ~100 us of one wave, and a handful of rows would cost the job that long.
// Every candidate that can be in row i's top-n passes a filter that is never tighter than the row's own running
// threshold (which only rises), and the final selection is by key (sum desc, column asc): the result is the row-major
// kernel's, bit for bit (tests/test_k3_cossim_gpu.py::test_symmetric_*).
//
// A job may come in row ranges (TFIDF.match enqueues four so that frame building overlaps the device): the ranges must
// ascend from row 0 without gaps; a range's rows are final after its own pass 1 (whoever pushes to them has a lower
// row number).  Anything else -- a range that does not continue the session, other matrices, top_n > 32, short lists,
// big to-sides (k3_lockstep.hip) -- runs the row-major kernel.
#include "k3_core.h"

#include <stdlib.h>

namespace pfz {

constexpr int kSymC = 2048;        // to-rows per block (the index is built with 2048-row blocks)
constexpr int kSymCap = 96;        // candidate keys per wave (ntop <= 32), as in the main kernel
constexpr int kSymKeep = 32;       // keys a row keeps between the passes at most
#ifndef PFZ_K3_SYM_F
#define PFZ_K3_SYM_F 128           // (tuning knobs of tools/build_variant.sh)
#endif
#ifndef PFZ_K3_SYM_PUSH
#define PFZ_K3_SYM_PUSH 512
#endif
constexpr int kSymF = PFZ_K3_SYM_F;         // staged foreign candidates per wave (flushed above kSymF - 64)
constexpr int kSymPush = PFZ_K3_SYM_PUSH;   // push slots per row; a row that is sent more is recomputed in full
constexpr int kSymMergeCap = kSymPush + 64;
constexpr int kSymSlices = 8;      // pass 2: a row that is recomputed in full is cut into at most this many slices of to-blocks ...
constexpr int kSymSlicedRows = 4096;   // ... for the first so many rows of the list (a single wave takes ~100 us for a whole row)

struct K3SymArgs {
    const int32_t *a_indptr;
    const int32_t *a_idx;
    const float *a_val;
    int32_t n;                // rows of the matrix == to-rows of the index
    const int32_t *tab;
    const int2 *post;
    int32_t nb, n_pieces, ntop, thr0;
    float scale, inv_scale;
    int32_t row_begin, row_end;   // the rows of this launch (modes 0 and 1, merge)
    int32_t *thrv;            // [n]           published thresholds (accept sum > thr)
    uint16_t *thr16;          // [nb * C]      their upper halves, in sweep order (see thr16_pos); rows >= n: 0x7f7f
    uint64_t *keys;           // [n][ntop]     a row's own candidates, sorted, 0 = none
    int32_t *push_cnt;        // [n]
    uint64_t *push_buf;       // [n][kSymPush] keys sum << 32 | ~(row that found it)
    int32_t *ovf;             // [1 + n]       ovf[0] = number of rows to recompute, then the rows
    uint64_t *part;           // [kSymSlicedRows][n_sl][ntop]  pass 2 in slices: the partial top-n of (row, slice of the to-blocks)
    int32_t ovf_base, ovf_max, n_sl;   // pass 2: listed rows [ovf_base, ovf_base + ovf_max), each cut into n_sl slices (1: whole rows -> result)
    int32_t *out_idx;
    float *out_val;
    int32_t exp;              // timing experiments (PFZ_K3_SYM_EXP, results wrong on purpose): 1 = no second filter, 2 = no threshold loads, 4 = no publishing
};

// where the upper half of row `row`'s threshold sits: a sweep step t of lane l reads the int4 slots i0 = 128 t + l and
// i0 + 64, i.e. the to-rows 512 t + 4 l + c and 512 t + 256 + 4 l + c of the block -- eight thresholds, one 16-byte load
__device__ inline int64_t thr16_pos(int row)
{
    const int b = row / kSymC, lr = row - b * kSymC;
    const int t = lr >> 9, rem = lr & 511, half = rem >> 8, q = rem & 255;
    return ((((int64_t)b * (kSymC / 512) + t) * 64 + (q >> 2)) * 8) + half * 4 + (q & 3);
}

typedef short short2v __attribute__((ext_vector_type(2)));

// sign bits of (upper half of sum) - (upper half of threshold) for two sums: clear = candidate
__device__ inline uint32_t upper_diff(int s_lo, int s_hi, int thr_pair)
{
    // {s_hi[31:16], s_lo[31:16]}: v_perm_b32 with selector 0x07060302 (bytes 3, 2 of the first operand over bytes 3, 2 of the second)
    const uint32_t p = __builtin_amdgcn_perm((uint32_t)s_hi, (uint32_t)s_lo, 0x07060302u);
    const short2v d = __builtin_bit_cast(short2v, p) - __builtin_bit_cast(short2v, (uint32_t)thr_pair);   // v_pk_sub_i16
    return __builtin_bit_cast(uint32_t, d);
}

__device__ inline void flush_foreign(uint64_t *fbuf, int &fcnt, int lane, const K3SymArgs &a, uint32_t inv_row)
{
    wave_sync();
    for (int e = lane; e < fcnt; e += 64) {
        const uint64_t en = fbuf[e];
        const int i = (int)(uint32_t)en;
        const int pos = atomicAdd(&a.push_cnt[i], 1);
        if (pos < kSymPush) a.push_buf[(int64_t)i * kSymPush + pos] = (en & 0xffffffff00000000ull) | inv_row;
    }
    wave_sync();
    fcnt = 0;
}

// the eight sums of one sweep step (to-rows j0 .. j0+3 and j1 .. j1+3) against the upper halves of those rows' thresholds
// (q: two per word, in that order) -- the rare path.  (Unrolled: with a rolled loop the compiler keeps the eight sums in
// scratch memory, stores in the hot path included.)
__device__ inline void foreign8(uint64_t *fbuf, int &fcnt, const int4 &v0, const int4 &v1, const int4 &q, int j0, int j1,
                                int lane, const K3SymArgs &a, uint32_t inv_row)
{
    const int xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    const int ws[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int x = xs[c];
        const int tt = (c & 1) ? (int)((uint32_t)ws[c >> 1] >> 16) : (ws[c >> 1] & 0xffff);
        const int j = (c < 4 ? j0 : j1) + (c & 3);
        // (conservative: sum > thr implies >= of the upper halves; never a zero or negative sum)
        const bool pred = (x >> 16) >= tt && x > a.thr0;
        const uint64_t mk = __ballot(pred);
        if (mk) {
            const int pos = fcnt + __popcll(mk & ((1ull << lane) - 1ull));
            if (pred) fbuf[pos] = ((uint64_t)(uint32_t)x << 32) | (uint32_t)j;
            fcnt += __popcll(mk);
            if (fcnt > kSymF - 64) flush_foreign(fbuf, fcnt, lane, a, inv_row);
        }
    }
}

// sweep_block of k3_core.h with the second filter.  qa / qb: the upper halves of the thresholds of the eight to-rows whose
// sums this lane reads in steps 0 / 1 (0x7fff.. where there is nothing to hand over: no sum's upper half reaches that --
// |sum| stays below 2^31 / 1.0001, pfz_cossim_topn_rows); those of steps 2 / 3 are loaded from tq_blk while steps 0 / 1
// are worked on.  hand_over = pass 1.  (Two rolled iterations of two steps, like the main kernel's sweep: the code of the
// rare paths exists twice, not four times.)
#ifndef PFZ_K3_SYM_TQ4
#define PFZ_K3_SYM_TQ4 0           // 1: the thresholds of all four sweep steps are loaded before the scatter (16 registers across it)
#endif
template <int N4, int kCap>
__device__ inline void sweep_block_sym(int4 *acc4, uint64_t *cand, TopState &st, int col0, int self_col, int ntop, int lane,
                                       int zero, int4 qa, int4 qb,
#if PFZ_K3_SYM_TQ4
                                       int4 qc, int4 qd,
#else
                                       const int4 *tq_blk,
#endif
                                       uint64_t *fbuf, int &fcnt, const K3SymArgs &a, uint32_t inv_row, bool hand_over)
{
    static_assert(N4 / 128 == 4, "four sweep steps per block");
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const int4 qq[2] = {qa, qb};
#if PFZ_K3_SYM_TQ4
        qa = qc;
        qb = qd;
#else
        if (h == 0 && hand_over) {
            qa = tq_blk[2 * 64 + lane];
            qb = tq_blk[3 * 64 + lane];
        }
#endif
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i0 = (2 * h + u) * 128 + lane, i1 = i0 + 64;
            const int4 v0 = acc4[i0], v1 = acc4[i1];
            acc4[i0] = make_int4(zero, zero, zero, zero);
            acc4[i1] = make_int4(zero, zero, zero, zero);
            const int mx = max3i(max3i(v0.x, v0.y, v0.z), max3i(v0.w, v1.x, v1.y), max3i(v1.z, v1.w, v1.w));
            const uint32_t d = upper_diff(v0.x, v0.y, qq[u].x) & upper_diff(v0.z, v0.w, qq[u].y) &
                               upper_diff(v1.x, v1.y, qq[u].z) & upper_diff(v1.z, v1.w, qq[u].w);
            const bool f = hand_over && (d & 0x80008000u) != 0x80008000u;       // some upper half reaches its row's
            const bool own = mx > st.thr;
            if (__ballot(own || f)) {
                if (__ballot(own)) {
                    push4<kCap>(cand, st, v0, col0 + i0 * 4, self_col, ntop, lane);
                    push4<kCap>(cand, st, v1, col0 + i1 * 4, self_col, ntop, lane);
                }
                if (__ballot(f)) foreign8(fbuf, fcnt, v0, v1, qq[u], col0 + i0 * 4, col0 + i1 * 4, lane, a, inv_row);
            }
        }
    }
}

// MODE: the pass (0: own block -> state; 1: the blocks above -> state + pushes; 2: all blocks of the listed rows -> result) --
// a template parameter so that every pass is a kernel of its own name in a trace and carries only its own code
template <int C, int MODE>
__global__ __launch_bounds__(64) void k3_sym_kernel(const K3SymArgs a)
{
    static_assert(C == kSymC, "thr16_pos() is written for 2048-row blocks");
    // accumulators first: they land at LDS address 0 and a posting's byte offset IS its LDS address (run_steps)
    __shared__ __attribute__((aligned(16))) struct {
        int acc[C];
        uint64_t cand[kSymCap];
        uint64_t fbuf[MODE == 1 ? kSymF : 1];      // (only pass 1 hands candidates over)
    } sm;
    int *const acc = sm.acc;
    uint64_t *const cand = sm.cand;
    uint64_t *const fbuf = sm.fbuf;
    int *const mark = (int *)(sm.cand + kSymCap) - 64;          // scatter scratch: the tail of the candidate buffer
    if ((uint32_t)(uintptr_t)sm.acc != 0u) __builtin_trap();    // layout assumption of run_steps()
    const int lane = threadIdx.x;
    int4 *acc4 = (int4 *)acc;
    constexpr int N4 = C / 4;
    constexpr int NT = N4 / 128;
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    for (int t = lane; t < C / 4; t += 64) acc4[t] = make_int4(0, 0, 0, 0);
    wave_sync();
    const char *post_bytes = (const char *)a.post;
    const int src4 = (4 * (lane & 15) + (lane >> 4)) * 4;
    const int sub8 = (lane & 15) * 8;
    const int dummy_addr = a.n_pieces << 7;
    const int nb = a.nb, ntop = a.ntop;
    constexpr int mode = MODE;
    const int4 *thr16q = (const int4 *)a.thr16;

    int n_items = a.row_end - a.row_begin;
    const int n_sl = mode == 2 ? a.n_sl : 1;
    const int per_sl = (nb + n_sl - 1) / n_sl;
    if (mode == 2) {
        int n_ovf = a.ovf[0];
        n_ovf = n_ovf > a.n ? a.n : n_ovf;
        const int hi = n_ovf < a.ovf_base + a.ovf_max ? n_ovf : a.ovf_base + a.ovf_max;
        n_items = hi > a.ovf_base ? (hi - a.ovf_base) * n_sl : 0;
    }
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int row = mode == 2 ? a.ovf[1 + a.ovf_base + item / n_sl] : a.row_begin + item;
        const int own = row / C;
        int b_lo = 0, b_hi = nb, b_first = own;
        if (mode == 2 && n_sl > 1) {
            b_lo = (item % n_sl) * per_sl;
            b_hi = b_lo + per_sl < nb ? b_lo + per_sl : nb;
            b_first = own >= b_lo && own < b_hi ? own : b_lo;
            if (b_lo >= b_hi) {       // (an empty trailing slice)
                if (lane < ntop) a.part[(int64_t)item * ntop + lane] = 0ull;
                continue;
            }
        }
        if (mode == 0) {
            b_lo = own;
            b_hi = own + 1;
        } else if (mode == 1) {
            b_lo = own + 1;
            b_first = b_lo;
            if (b_lo >= nb) continue;      // (the launch does not cover the last block's rows; kept for safety)
        }
        const int p0 = a.a_indptr[row], p1 = a.a_indptr[row + 1];
        const int nnz = p1 - p0;
        const int self_col = mode == 1 ? -1 : row;      // above the own block there is no diagonal
        const uint32_t inv_row = ~(uint32_t)row;
        TopState st;
        st.cnt = 0;
        st.thr = a.thr0;
        st.pushed = 0;
        bool warmed = false;
        if (mode == 1) {
            // the state of pass 0: threshold and the sorted keys (zeros at the end).  The marker scratch of the scatter is the
            // LAST 64 ints of cand (keys 64..95): the kept keys (< 32) are out of its way
            st.thr = a.thrv[row];
            const uint64_t k = lane < ntop ? a.keys[(int64_t)row * ntop + lane] : 0ull;
            st.cnt = __popcll(__ballot(k != 0ull));
            if (k) cand[lane] = k;
            warmed = true;
            wave_sync();
        }
        int pub = st.thr;
        int fcnt = 0;

        const int n_blk = b_hi - b_lo;
        int cur0 = 0, nxt0 = 0;
        float as0 = 0.f;
        const bool have0 = lane < nnz;
        const int32_t *trow = a.tab;
        if (have0) {
            as0 = a.a_val[p0 + lane] * a.scale;
            trow = a.tab + (int64_t)a.a_idx[p0 + lane] * nb;
            cur0 = trow[b_first];
            nxt0 = trow[b_first + 1];
        }

        for (int it = 0, b = b_first; it < n_blk; ++it) {
            const int s = cur0, e = have0 ? nxt0 : cur0;
            const int b_next = b + 1 < b_hi ? b + 1 : b_lo;
            // the thresholds of the block's first two sweep steps, in flight across the scatter (the other two follow inside the sweep)
            const int4 *tq_blk = thr16q + (int64_t)b * NT * 64;
            int4 qa = make_int4(0x7fff7fff, 0x7fff7fff, 0x7fff7fff, 0x7fff7fff), qb = qa;
#if PFZ_K3_SYM_TQ4
            int4 qc = qa, qd = qa;
#endif
            if (mode == 1 && !(a.exp & 2)) {
                qa = tq_blk[lane];
                qb = tq_blk[64 + lane];
#if PFZ_K3_SYM_TQ4
                qc = tq_blk[128 + lane];
                qd = tq_blk[192 + lane];
#endif
            }
            bool touched = __ballot(e > s) != 0;
            if (touched) scatter_pieces(acc, post_bytes, mark, e - s, s, as0, lane, src4, sub8, dummy_addr);
            if (have0 && it + 1 < n_blk) {
                cur0 = trow[b_next];
                nxt0 = trow[b_next + 1];
            }
            for (int c0 = p0 + 64; c0 < p1; c0 += 64) {  // rows with more than 64 n-grams
                int s2 = 0, e2 = 0;
                float as2 = 0.f;
                if (c0 + lane < p1) {
                    const int k = a.a_idx[c0 + lane];
                    as2 = a.a_val[c0 + lane] * a.scale;
                    s2 = a.tab[(int64_t)k * nb + b];
                    e2 = a.tab[(int64_t)k * nb + b + 1];
                }
                if (__ballot(e2 > s2)) {
                    touched = true;
                    scatter_pieces(acc, post_bytes, mark, e2 - s2, s2, as2, lane, src4, sub8, dummy_addr);
                }
            }
            if (touched) {
                wave_sync();
                if (!warmed) {
                    warmed = true;
                    if (ntop <= kWarmMaxTop) {
                        const int t = warm_threshold<N4>(acc4, 0, ntop + 1, lane);
                        st.thr = t > st.thr ? t : st.thr;
                    }
                }
#if PFZ_K3_SYM_TQ4
                sweep_block_sym<N4, kSymCap>(acc4, cand, st, b * C, self_col, ntop, lane, zero, qa, qb, qc, qd, fbuf, fcnt, a, inv_row, mode == 1 && !(a.exp & 1));
#else
                sweep_block_sym<N4, kSymCap>(acc4, cand, st, b * C, self_col, ntop, lane, zero, qa, qb, tq_blk, fbuf, fcnt, a, inv_row, mode == 1 && !(a.exp & 1));
#endif
                wave_sync();
                if (mode == 1 && st.thr > pub && !(a.exp & 4)) {        // tell the rows below: fewer of their sums are candidates of this row
                    pub = st.thr;
                    if (lane == 0) {
                        a.thrv[row] = pub;
                        a.thr16[thr16_pos(row)] = (uint16_t)((uint32_t)pub >> 16);
                    }
                }
            }
            b = b_next;
        }

        if (fcnt) flush_foreign(fbuf, fcnt, lane, a, inv_row);
        compact<kSymCap>(cand, st, ntop, lane);
        if (mode == 2 && n_sl > 1) {
            if (lane < ntop) a.part[(int64_t)item * ntop + lane] = lane < st.cnt ? cand[lane] : 0ull;
        } else if (mode == 2) {
            for (int r = lane; r < ntop; r += 64) {
                const uint64_t key = r < st.cnt ? cand[r] : 0ull;
                a.out_idx[(int64_t)row * ntop + r] = key ? (int32_t)(~(uint32_t)key) : -1;
                a.out_val[(int64_t)row * ntop + r] = key ? (float)(int32_t)(uint32_t)(key >> 32) * a.inv_scale : 0.f;
            }
        } else {
            if (lane < ntop) a.keys[(int64_t)row * ntop + lane] = lane < st.cnt ? cand[lane] : 0ull;
            if (lane == 0 && (mode == 0 || st.thr > pub)) {
                a.thrv[row] = st.thr;
                a.thr16[thr16_pos(row)] = (uint16_t)((uint32_t)st.thr >> 16);
            }
        }
        wave_sync();    // cand is reused by the next item
    }
}

// own keys + pushed keys -> the row's sorted top-n (one wave per row); rows that were sent more than kSymPush are listed
__global__ __launch_bounds__(256) void k3_sym_merge(const K3SymArgs a)
{
    __shared__ __attribute__((aligned(16))) uint64_t cand_all[4][kSymMergeCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = a.row_begin + blockIdx.x * 4 + wave;
    if (row >= a.row_end) return;
    uint64_t *cand = cand_all[wave];
    const int ntop = a.ntop;
    const int pushed = a.push_cnt[row];
    if (pushed > kSymPush) {
        if (lane == 0) {
            const int p = atomicAdd(&a.ovf[0], 1);
            if (p < a.n) a.ovf[1 + p] = row;
        }
        return;
    }
    TopState st;
    st.cnt = 0;
    st.thr = 0;
    st.pushed = 0;
    const uint64_t k = lane < ntop ? a.keys[(int64_t)row * ntop + lane] : 0ull;
    if (pushed > 0) {
        const uint64_t mk = __ballot(k != 0ull);
        if (k) cand[__popcll(mk & ((1ull << lane) - 1ull))] = k;
        st.cnt = __popcll(mk);
        for (int e = lane; e < pushed; e += 64) cand[st.cnt + e] = a.push_buf[(int64_t)row * kSymPush + e];
        st.cnt += pushed;
        // (the capacity is the number of keys a lane holds in registers: most rows were sent a few dozen candidates)
        if (st.cnt <= 64) compact<64>(cand, st, ntop, lane);
        else if (st.cnt <= 256) compact<256>(cand, st, ntop, lane);
        else compact<kSymMergeCap>(cand, st, ntop, lane);
    } else {
        if (lane < ntop) cand[lane] = k;       // sorted already, zeros at the end
        st.cnt = ntop;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (int r = lane; r < ntop; r += 64) {
        const uint64_t key = r < st.cnt ? cand[r] : 0ull;
        a.out_idx[(int64_t)row * ntop + r] = key ? (int32_t)(~(uint32_t)key) : -1;
        a.out_val[(int64_t)row * ntop + r] = key ? (float)(int32_t)(uint32_t)(key >> 32) * a.inv_scale : 0.f;
    }
}

// pass 2 in slices: the n_sl partial top-n lists of every recomputed row -> its result (one wave per row)
__global__ __launch_bounds__(256) void k3_sym_merge_slices(const K3SymArgs a)
{
    __shared__ __attribute__((aligned(16))) uint64_t cand_all[4][256];
    static_assert(kSymSlices * kSymKeep <= 256, "the partial lists of a row fit one 256-key compaction");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t *cand = cand_all[wave];
    const int ntop = a.ntop, n_sl = a.n_sl;
    int n_ovf = a.ovf[0];
    n_ovf = n_ovf > a.n ? a.n : n_ovf;
    const int rows = n_ovf < a.ovf_max ? n_ovf : a.ovf_max;      // (ovf_base == 0: the sliced rows are the first of the list)
    for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
        const int row = a.ovf[1 + r];
        TopState st;
        st.cnt = 0;
        st.thr = 0;
        st.pushed = 0;
        const int total = n_sl * ntop;
        for (int e0 = 0; e0 < total; e0 += 64) {
            const uint64_t k = e0 + lane < total ? a.part[(int64_t)r * total + e0 + lane] : 0ull;
            const uint64_t mk = __ballot(k != 0ull);
            if (k) cand[st.cnt + __popcll(mk & ((1ull << lane) - 1ull))] = k;
            st.cnt += __popcll(mk);
        }
        compact<256>(cand, st, ntop, lane);
        for (int q = lane; q < ntop; q += 64) {
            const uint64_t key = q < st.cnt ? cand[q] : 0ull;
            a.out_idx[(int64_t)row * ntop + q] = key ? (int32_t)(~(uint32_t)key) : -1;
            a.out_val[(int64_t)row * ntop + q] = key ? (float)(int32_t)(uint32_t)(key >> 32) * a.inv_scale : 0.f;
        }
        wave_sync();
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------

struct K3SymState {
    pfz_ctx *ctx = nullptr;
    int64_t n = 0;
    int32_t *thrv = nullptr;
    uint16_t *thr16 = nullptr;
    uint64_t *keys = nullptr;
    int32_t *push_cnt = nullptr;
    uint64_t *push_buf = nullptr;
    int32_t *ovf = nullptr;
    uint64_t *part = nullptr;
    // the running session: the next range must start where the last one ended, with the same job
    int64_t next_row = -1;
    uint64_t a_serial = 0;
    const pfz_topn *out = nullptr;
    int32_t ntop = 0, thr0 = 0;
    float scale = 0.f;
    int64_t launches = 0, rows = 0;      // pfz_index_symmetric_launches
};

void k3_sym_free(pfz_index *ix)
{
    K3SymState *s = ix->sym;
    if (!s) return;
    if (s->thrv) pool_free(s->thrv);
    if (s->thr16) pool_free(s->thr16);
    if (s->keys) pool_free(s->keys);
    if (s->push_cnt) pool_free(s->push_cnt);
    if (s->push_buf) pool_free(s->push_buf);
    if (s->ovf) pool_free(s->ovf);
    if (s->part) pool_free(s->part);
    delete s;
    ix->sym = nullptr;
}

static int sym_env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// 1: start a session with this range, 2: this range continues the running session, 0: not a job for this form
int k3_sym_wanted(const pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int64_t row_begin, int64_t row_end, int32_t ntop,
                  int32_t thr0, float scale, int32_t exclude_diag, int64_t diag_offset, const pfz_topn *out)
{
    (void)ctx;
    const int force = sym_env_int("PFZ_K3_SYM", -1);    // 0: never; 1: whenever the arithmetic allows (tests); default: auto
    if (force == 0) return 0;
    if (!exclude_diag || diag_offset != 0 || A->serial != ix->src_serial || A->n_rows != ix->n_rows) return 0;
    if (ix->block_cols != kSymC || ntop > kSymKeep || ix->n_blocks < 2 || ix->n_rows >= ((int64_t)1 << 30)) return 0;
    const K3SymState *s = ix->sym;
    if (row_begin > 0) {
        const bool cont = s && s->next_row == row_begin && s->a_serial == A->serial && s->out == out && s->ntop == ntop &&
                          s->thr0 == thr0 && s->scale == scale;
        return cont ? 2 : 0;
    }
    if (force == 1) return 1;
    // auto: where halving K3 pays for five more launches and the state round trip, and where the row-major kernel is the one that
    // would run (k3_lockstep.hip takes the to-sides beyond 250 000 rows); a first range of less than a fifth of the rows is a
    // shard of a bigger job (bench --scaling strong), not the start of a whole self-match
    if (ix->n_rows < sym_env_int("PFZ_K3_SYM_MIN", 20480) || ix->n_rows > 250000) return 0;
    return (row_end - row_begin) * 5 >= ix->n_rows ? 1 : 0;
}

int k3_sym_launch(pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int64_t row_begin, int64_t row_end, int32_t ntop,
                  int32_t thr0, float scale, float inv_scale, pfz_topn *out, bool start)
{
    const int64_t n = ix->n_rows;
    const int nb = ix->n_blocks;
    K3SymState *s = ix->sym;
    if (!s) {
        s = new K3SymState();
        s->ctx = ctx;
        s->n = n;
        ix->sym = s;      // (freed with the index, whatever happens below)
        PFZ_TRY(pool_alloc(ctx, &s->thrv, (size_t)n * sizeof(int32_t)));
        PFZ_TRY(pool_alloc(ctx, &s->thr16, (size_t)nb * kSymC * sizeof(uint16_t)));
        PFZ_TRY(pool_alloc(ctx, &s->keys, (size_t)n * kSymKeep * sizeof(uint64_t)));
        PFZ_TRY(pool_alloc(ctx, &s->push_cnt, (size_t)n * sizeof(int32_t)));
        PFZ_TRY(pool_alloc(ctx, &s->push_buf, (size_t)n * kSymPush * sizeof(uint64_t)));
        PFZ_TRY(pool_alloc(ctx, &s->ovf, (size_t)(n + 1) * sizeof(int32_t)));
        PFZ_TRY(pool_alloc(ctx, &s->part, (size_t)kSymSlicedRows * kSymSlices * kSymKeep * sizeof(uint64_t)));
    }
    if (!s->thrv || !s->thr16 || !s->keys || !s->push_cnt || !s->push_buf || !s->ovf || !s->part) {
        set_error("pfz_cossim_topn (symmetric): the session buffers of this index could not be allocated earlier");
        return PFZ_ERR_INVALID;
    }
    s->next_row = -1;     // (no session while this call can still fail)
    K3SymArgs a;
    a.a_indptr = A->indptr;
    a.a_idx = A->indices;
    a.a_val = A->data;
    a.n = (int32_t)n;
    a.tab = ix->tab;
    a.post = ix->post;
    a.nb = nb;
    a.n_pieces = ix->n_pieces;
    a.ntop = ntop;
    a.thr0 = thr0;
    a.scale = scale;
    a.inv_scale = inv_scale;
    a.thrv = s->thrv;
    a.thr16 = s->thr16;
    a.keys = s->keys;
    a.push_cnt = s->push_cnt;
    a.push_buf = s->push_buf;
    a.ovf = s->ovf;
    a.part = s->part;
    a.ovf_base = 0;
    a.ovf_max = 0;
    a.n_sl = 1;
    a.out_idx = out->idx;
    a.out_val = out->val;
    a.exp = sym_env_int("PFZ_K3_SYM_EXP", 0);
    if (start) {
        // pass 0 over ALL rows: every row's first threshold is there before anybody hands anything over
        PFZ_HIP(hipMemsetAsync(s->push_cnt, 0, (size_t)n * sizeof(int32_t), ctx->stream));
        PFZ_HIP(hipMemsetAsync(s->thr16, 0x7f, (size_t)nb * kSymC * sizeof(uint16_t), ctx->stream));
        a.row_begin = 0;
        a.row_end = (int32_t)n;
        hipLaunchKernelGGL((k3_sym_kernel<kSymC, 0>), dim3((unsigned)n), dim3(64), 0, ctx->stream, a);
    }
    PFZ_HIP(hipMemsetAsync(s->ovf, 0, sizeof(int32_t), ctx->stream));
    // pass 1: the rows of this range that have blocks above their own
    const int64_t last_block_row = (int64_t)(nb - 1) * kSymC;
    a.row_begin = (int32_t)row_begin;
    a.row_end = (int32_t)(row_end < last_block_row ? row_end : last_block_row);
    if (a.row_end > a.row_begin)
        hipLaunchKernelGGL((k3_sym_kernel<kSymC, 1>), dim3((unsigned)(a.row_end - a.row_begin)), dim3(64), 0, ctx->stream, a);
    // merge, then the rows that were sent too much
    a.row_begin = (int32_t)row_begin;
    a.row_end = (int32_t)row_end;
    hipLaunchKernelGGL(k3_sym_merge, dim3((unsigned)((row_end - row_begin + 3) / 4)), dim3(256), 0, ctx->stream, a);
    // pass 2: the first kSymSlicedRows of the listed rows in slices of the to-blocks (a whole row is ~100 us of one wave: a handful
    // of rows would cost that much wall time), their partial lists merged; whatever is listed beyond, as whole rows
    {
        const unsigned grid2 = (unsigned)ctx->prop.multiProcessorCount * 16;
        const int per = (nb + kSymSlices - 1) / kSymSlices;
        a.n_sl = (nb + per - 1) / per;
        a.ovf_base = 0;
        a.ovf_max = kSymSlicedRows;
        if (a.n_sl > 1) {
            hipLaunchKernelGGL((k3_sym_kernel<kSymC, 2>), dim3(grid2), dim3(64), 0, ctx->stream, a);
            hipLaunchKernelGGL(k3_sym_merge_slices, dim3(256), dim3(256), 0, ctx->stream, a);
            a.ovf_base = kSymSlicedRows;
        }
        a.n_sl = 1;
        a.ovf_max = (int32_t)n;
        hipLaunchKernelGGL((k3_sym_kernel<kSymC, 2>), dim3(grid2), dim3(64), 0, ctx->stream, a);
    }
    PFZ_HIP(hipGetLastError());
    s->next_row = row_end;
    s->launches += 1;
    s->rows += row_end - row_begin;
    s->a_serial = A->serial;
    s->out = out;
    s->ntop = ntop;
    s->thr0 = thr0;
    s->scale = scale;
    return PFZ_OK;
}

}  // namespace pfz

extern "C" int pfz_index_symmetric_launches(const pfz_index *ix, int64_t *launches, int64_t *rows)
{
    PFZ_REQUIRE(ix, "pfz_index_symmetric_launches: NULL index");
    if (launches) *launches = ix->sym ? ix->sym->launches : 0;
    if (rows) *rows = ix->sym ? ix->sym->rows : 0;
    return PFZ_OK;
}
