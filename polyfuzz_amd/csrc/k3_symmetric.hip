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
// How.  Three passes of one kernel template (k3_sym_kernel<C, MODE>), a re-deal of the accumulator slots, and a merge:
//   0  every row x its OWN block (both directions of a pair inside a block are computed: no exchange there) -> the
//      row's first top-n and threshold, written to HBM (keys[row][ntop], thrv[row]).  On a sorted list (the reference's
//      company names are) a row's best matches sit next to it: the thresholds are high from the start.
//   order (round 5)  The second filter of pass 1 -- "is this sum a candidate for the row of the CELL?" -- used to test
//      every sum against its own row's threshold (eight 16-bit thresholds per lane and sweep step: 11 vector
//      instructions and 16 bytes of loads per step, a quarter of pass 1).  Which accumulator slot a to-row's sums land
//      in is free, so after pass 0 every block's rows are RE-DEALT TO THE SLOTS BY THRESHOLD (k3_sym_order): the eight
//      slots a lane reads in one sweep step hold rows of neighbouring thresholds, and one comparison of the lane's
//      maximum against the minimum of those eight (gmin) decides the step.  The re-deal keeps every row in its LDS bank
//      (slot = row mod 32): row r of bank class rho = r mod 32 gets rank i among the 64 rows of its class, the eight
//      lanes l = rho / 4 + 8 k of the four sweep steps own the class' 64 slots, ranks 2 p and 2 p + 1 go to (step p / 8,
//      lane rho / 4 + 8 (p mod 8)) -- so the bank order of the heavy lists (k_index_bank_order) survives, and the
//      re-dealt index is the old one with the postings' slot fields rewritten (k3_sym_repost: one pass over a copy).
//      CPU simulation on the 100 000 names (tools/sim_sym_slot_order.py): a step enters the rare path in 25 % of the
//      cases (exact per-row test: 15 %; groups of eight rows as they lie: 54 %).
//   1  row j x the blocks ABOVE its own: state restored, scatter as ever (on the re-dealt postings); the sweep tests the
//      maximum of a lane's eight sums against min(the row's own threshold, gmin of the lane's eight slots) -- two vector
//      instructions, 4 bytes of loads per lane and step, in flight across the scatter.  The rare path loads the eight
//      rows of the lane's slots (inv8) and their exact thresholds (thr8), keeps what beats the row's own threshold
//      (key = sum, true column) and puts what beats the CELL's row's threshold into a 128-entry LDS buffer as (sum, row
//      of the cell); the buffer is flushed -- one returning atomic per entry on push_cnt[i], one store into
//      push_buf[i] -- when it is half full and at the end of the row: about once per row (23 candidates per row on
//      the 100 000 company names).
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

#include <stdio.h>
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
#ifndef PFZ_K3_SYM_SIDE_STREAMS
#define PFZ_K3_SYM_SIDE_STREAMS 2      // (tuning builds, tools/build_variant.sh: 1 = every range's chain on one side stream, round 6's first form; up to 4)
#endif
constexpr int kSymSides = PFZ_K3_SYM_SIDE_STREAMS;      // side streams of a streamed session (k3_sym_launch_streamed)
static_assert(kSymSides >= 1 && kSymSides <= 4, "one to four side streams");
constexpr int kSymSlices = 8;      // pass 2: a row that is recomputed in full is cut into at most this many slices of to-blocks ...
constexpr int kSymSlicedRows = 4096;   // ... for the first so many rows of the list (a single wave takes ~100 us for a whole row)
#ifndef PFZ_K3_SYM_EXP
#define PFZ_K3_SYM_EXP 0           // timing experiments (tools/build_variant.sh -DPFZ_K3_SYM_EXP=n: results wrong on purpose); 0 = the product
#endif
constexpr int kSymMag = 1;          // "magnet" rows per LDS bank class and block: the rows of the lowest thresholds are taken out of the hand-over
constexpr int kSymMagBlocks = 8;   // ... and walk the blocks below their own themselves, in items of so many to-blocks
constexpr int kSymP0PerCu = 128;    // pass 0: persistent workgroups per CU (~3 rows each at 100 000 rows; 18 / 36 / 72 / 144 per CU: 1.842 / 1.780 / 1.758 / 1.757 ms of K3, round 5)
constexpr int kSymDoneStride = 32;  // a block's item counter (streamed sessions) has a 128-byte line of its own: 2 000 atomics each, from every XCD
constexpr int kNoThr = 0x7fffffff;     // threshold of a slot without a row (the last block's tail): no sum reaches it

struct K3SymArgs {
    const int32_t *a_indptr;
    const int32_t *a_idx;
    const float *a_val;
    int32_t n;                // rows of the matrix == to-rows of the index
    const int32_t *tab;
    const int2 *post;         // the index' postings: passes 0 and 2
    const int2 *post_sym;     // the same pieces with the slot fields re-dealt by threshold: pass 1
    const uint16_t *pblk;     // to-block of every piece
    const int32_t *n_pieces1; // (device) pieces of the index, the dummy piece 0 included
    int32_t nb, ntop, thr0;
    float scale, inv_scale;
    int32_t row_begin, row_end;   // the rows of this launch (modes 0 and 1, merge)
    int32_t n_parts, my_part, per;   // the job cut over n_parts GPUs (k3_sym_sharded): this part works on the rows = my_part (mod n_parts); per = ceil(n / n_parts)
    int32_t thr_by_part;          // thrv is laid out part by part (k3_sym_sharded) -- or by row
    uint64_t *keys_out;           // != NULL: the merge / pass 2 leave every row's sorted keys here ([n][ntop]) instead of (index, score)
    int32_t *thrv;            // [n]              a row's threshold after pass 0 (accept sum > thr)
    uint16_t *slot4;          // [nb * C]         4 * (slot of to-row b * C + r): its accumulator's byte offset in pass 1
    uint32_t *gmin;           // [nb][64]         (block, lane) -> byte t: the top byte of the minimum threshold of the eight slots the lane reads in sweep step t (0xff: none)
    int32_t *thr_slot;        // [nb * C]         cell (block * C + slot) -> the threshold of the row that owns the slot
    uint16_t *row_slot;       // [nb * C]         ... and that row (inside the block)
    int32_t *mag;             // [nb][32 * kSymMag]  the magnet rows of every block (-1: none)
    int32_t n_mag_items, mag_b0, mag_row_end;   // pass 1: the first so many items are magnet items (block mag_b0 + ..., class, slice of the lower blocks), for the magnets below mag_row_end
    uint64_t *keys;           // [n][ntop]        a row's own candidates, sorted, 0 = none
    int32_t *push_cnt;        // [n]
    uint64_t *push_buf;       // [n][kSymPush]    keys sum << 32 | ~(row that found it)
    int32_t *ovf;             // [1 + n]          ovf[0] = number of rows to recompute, then the rows
    uint64_t *part;           // [kSymSlicedRows][n_sl][ntop]  pass 2 in slices: the partial top-n of (row, slice of the to-blocks)
    int32_t ovf_base, ovf_max, n_sl;   // pass 2: listed rows [ovf_base, ovf_base + ovf_max), each cut into n_sl slices (1: whole rows -> result)
    int32_t *out_idx;
    float *out_val;
    int32_t *host_idx;        // streamed session: a mirror of the result in pinned HOST memory (the frame builder reads it there: no
    float *host_val;          //   device-to-host copy, no stream for the host to wait on), or NULL
    uint32_t *done;           // [nb] streamed session (k3_sym_launch_streamed): pass-1 items finished per block of their row, or NULL
    int32_t blk_lo, blk_hi;   // k3_sym_wait: the blocks whose items must have finished
};

// Where a row's pass-0 threshold sits in thrv: by row -- or, when the job is cut over n_parts GPUs, part by part (part p's
// rows p, p + n_parts, ... are consecutive), so that an in-place all-gather of the parts' stretches completes the array.
__device__ inline int thr_pos(const K3SymArgs &a, int row)
{
    return a.thr_by_part ? (row % a.n_parts) * a.per + row / a.n_parts : row;
}
__device__ inline bool row_is_mine(const K3SymArgs &a, int row) { return a.n_parts == 1 || row % a.n_parts == a.my_part; }

// ---- the re-deal of a block's rows to the accumulator slots (after pass 0) -------------------------------------------

// slot of the row with rank i (0 = lowest threshold) among the 64 rows of bank class rho of its block; *t / *l: the sweep
// step and the lane that read the slot, *e: its place among the lane's eight sums (v0.x .. v0.w, v1.x .. v1.w)
__host__ __device__ inline int sym_slot(int rho, int i, int *t, int *l, int *e)
{
    const int p = i >> 1, half = i & 1, c = rho & 3;
    *t = p >> 3;
    *l = (rho >> 2) + 8 * (p & 7);
    *e = half * 4 + c;
    return 512 * *t + 256 * half + 4 * *l + c;       // = rho (mod 32): the row stays in its LDS bank
}

// one workgroup per to-block: thresholds of pass 0 -> slot4, gmin, thr_slot, row_slot.  A wave ranks two bank classes (lane q =
// row rho + 32 q) by counting: rank = number of rows of the class with a smaller (threshold, q).
__global__ __launch_bounds__(1024) void k3_sym_order(const K3SymArgs a)
{
    __shared__ int s_gmin[256];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 256) s_gmin[threadIdx.x] = kNoThr;
    if (threadIdx.x == 0 && a.done) a.done[b * kSymDoneStride] = 0u;
    __syncthreads();
    for (int rho = wave * 2; rho < wave * 2 + 2; ++rho) {
        const int r = rho + 32 * lane;
        const int row = b * kSymC + r;
        // (the chores that were memsets of their own in front of pass 0, ~6 us each on the stream: nobody has pushed anything to a
        // row yet -- pass 0 hands nothing over --, and no pass-1 item has counted itself)
        if (row < a.n) a.push_cnt[row] = 0;
        const int thr = row < a.n ? a.thrv[thr_pos(a, row)] : kNoThr;
        // (thresholds are >= 0: they start at thr0 >= 0 and only rise)
        const uint64_t key = ((uint64_t)(uint32_t)thr << 6) | (uint32_t)lane;
        int rank = 0;
#pragma unroll 16
        for (int m = 0; m < 64; ++m) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, m);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), m);
            rank += ((((uint64_t)hi << 32) | lo) < key) ? 1 : 0;
        }
        int t, l, e;
        const int slot = sym_slot(rho, rank, &t, &l, &e);
        // Magnets: the row(s) of the lowest threshold of every bank class.  A low threshold draws candidates from every row
        // below (93 per row for the lowest of a class, 0.3 for the highest: CPU census of the 100 000 names) and its group's
        // minimum sends a quarter of all sweep steps into the rare path; such a row is sent nothing -- no sum reaches kNoThr --
        // and fetches its matches in the blocks below its own itself (magnet items of pass 1).  Block 0 has nothing below.
        const bool magnet = rank < kSymMag && row < a.n && b > 0;
        if (rank < kSymMag) a.mag[(int64_t)b * (32 * kSymMag) + rho * kSymMag + rank] = magnet ? row : -1;
        a.slot4[(int64_t)b * kSymC + r] = (uint16_t)(slot * 4);
        a.thr_slot[(int64_t)b * kSymC + slot] = magnet ? kNoThr : thr;
        a.row_slot[(int64_t)b * kSymC + slot] = (uint16_t)r;
        if (!magnet) atomicMin(&s_gmin[t * 64 + l], thr);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        // what the sweep compares is the TOP BYTE of sums and minima (one v_cmp_ge_u32_sdwa, four minima per dword: the 16 bytes
        // of exact minima per lane and block cost pass 1 a tenth of its time, measured); conservative -- the exact thresholds
        // decide when the stage is drained -- and coarse only where it does not matter: 20.0 -> 21.5 % of the sweep steps
        // enter the rare path (CPU simulation)
        uint32_t w = 0;
        for (int t = 0; t < 4; ++t) {
            const int g = s_gmin[t * 64 + threadIdx.x];
            w |= (g == kNoThr ? 0xffu : (uint32_t)g >> 24) << (8 * t);
        }
        a.gmin[(int64_t)b * 64 + threadIdx.x] = w;
    }
}

// post -> post_sym: every posting's slot field through its block's slot4 (a thread takes two postings = 16 bytes; padding
// entries keep their value 0 and get the slot of whatever row they named: any slot will do for them)
__global__ __launch_bounds__(256) void k3_sym_repost(const K3SymArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // pair of postings
    if (i >= (int64_t)*a.n_pieces1 * (kPiece / 2)) return;        // (the grid covers the index' piece CAPACITY)
    const int piece = (int)(i >> 3);
    const int b = a.pblk[piece];
    int4 e = ((const int4 *)a.post)[i];
    const uint16_t *s4 = a.slot4 + (int64_t)b * kSymC;
    if (piece > 0) {                    // (the all-zero dummy, piece 0, stays as it is)
        e.x = s4[e.x >> 2];
        e.z = s4[e.z >> 2];
    }
    ((int4 *)a.post_sym)[i] = e;
}

// ---- pass 1's hand-over ------------------------------------------------------------------------------------------------

// Staged candidates of pass 1: sum << 32 | flags << 30 | cell, cell = block * C + slot (n < 2^30).  The sweep knows a cell by
// its SLOT; which row owns the slot (row_slot) and that row's exact threshold (thr_slot) are looked up when the stage is
// drained -- 64 entries per round trip instead of a dependent load in the sweep (measured, first version of the re-dealt
// slots: the loads in the sweep's rare path cost as much as the per-row filter they replaced).
constexpr uint32_t kStageOwn = 1u << 30;     // the sum beat the from-row's own threshold: a candidate of the from-row
constexpr uint32_t kStageFgn = 1u << 31;     // the sum beat the minimum threshold of its lane's eight slots: maybe a candidate of the cell's row

// ---- a session in ONE pass-1 launch whose row ranges are handed on as they finish (k3_sym_launch_streamed) --------------------
// Whatever pass 1 leaves for the merge -- a row's keys, the candidates pushed to other rows -- is stored WRITE-THROUGH at agent
// scope (sc1: no dirty line stays behind in this XCD's L2), and an item counts itself on `done[block of its row]` only after
// those stores are acknowledged (s_waitcnt vmcnt(0); k3_lockstep.hip's chunk flags work the same way).  The merge of a row range
// is a kernel of its own on a side stream, behind k3_sym_wait: its start invalidates what its XCD's L2 may hold of these rows,
// and every row of the range -- and every row that pushes to one, all of them in lower or the same blocks -- has counted.
#ifndef PFZ_K3_SYM_PLAIN_STORES
#define PFZ_K3_SYM_PLAIN_STORES 0      // 1: A/B builds only (tools/build_variant.sh): plain stores -- a streamed session is then NOT coherent
#endif
__device__ inline void store_coherent(uint64_t *p, uint64_t v)
{
#if PFZ_K3_SYM_PLAIN_STORES
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

__device__ inline void sym_item_done(const K3SymArgs &a, int blk, int lane)
{
    if (!a.done) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&a.done[blk * kSymDoneStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// items of pass 1 that belong to block b of a whole job in one launch: its rows (the last block's have nothing above) + the magnet
// items of its 32 * kSymMag magnet slots (items of empty slots and of slices that lie at or above the block count too)
__device__ inline uint32_t sym_items_of_block(const K3SymArgs &a, int b)
{
    const int rows = b < a.nb - 1 ? (a.n - b * kSymC < kSymC ? a.n - b * kSymC : kSymC) : 0;
    return (uint32_t)(rows + 32 * kSymMag * ((a.nb - 1 + kSymMagBlocks - 1) / kSymMagBlocks));
}

// one wave on the side stream: returns when every pass-1 item of the blocks [blk_lo, blk_hi) has counted itself.  A count that
// does not arrive within ~10 s is a bug of this file: trap (a loud HIP error) rather than a stream that never ends.
// (It also does the two chores that would otherwise be launches of their own, each ~20 us of host time on the side stream: it
// announces the range BEFORE -- whose kernels have finished, the stream is in order -- through its word in pinned host memory, and it
// clears the list of rows to recompute for the merge that follows.  flag == NULL / blk_lo == blk_hi: nothing to announce / to wait for.)
__global__ __launch_bounds__(64) void k3_sym_wait(const K3SymArgs a, int32_t *flag, int32_t flag_value)
{
    const int lane = threadIdx.x;
    if (lane == 0) {
        if (flag) __hip_atomic_store(flag, flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.blk_lo < a.blk_hi) a.ovf[0] = 0;       // (the last launch only announces: the session's buffers are not touched any more)
    }
    const uint64_t t0 = wall_clock64();
    for (int b0 = a.blk_lo; b0 < a.blk_hi; b0 += 64) {
        const int b = b0 + lane;
        const bool on = b < a.blk_hi;
        const uint32_t want = on ? sym_items_of_block(a, b) : 0u;
        for (;;) {
            const uint32_t got = on ? __hip_atomic_load(&a.done[b * kSymDoneStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (!__ballot(got < want)) break;
            if (wall_clock64() - t0 > 1000000000ull) __builtin_trap();      // (100 MHz)
            __builtin_amdgcn_s_sleep(32);
        }
    }
}

template <int kCap>
__device__ inline void drain_stage(uint64_t *cand, TopState &st, uint64_t *fbuf, int &fcnt, int ntop, int lane,
                                   const K3SymArgs &a, uint32_t inv_row)
{
    wave_sync();
    for (int e0 = 0; e0 < fcnt; e0 += 64) {
        const bool on = e0 + lane < fcnt;
        const uint64_t en = on ? fbuf[e0 + lane] : 0ull;
        const uint32_t lo = (uint32_t)en;
        const int cell = (int)(lo & 0x3fffffffu);
        const int x = (int)(uint32_t)(en >> 32);
        int rowl = 0, thr = kNoThr;
        if (on) {
            rowl = a.row_slot[cell];
            thr = a.thr_slot[cell];
        }
        const int row = (cell & ~(kSymC - 1)) + rowl;
#if PFZ_K3_SYM_EXP == 5      // (what-if 5: staged and drained, but nothing is pushed)
        if (false) {
#else
        if ((lo & kStageFgn) && x > thr) {         // one returning atomic, one store
#endif
            const int pos = atomicAdd(&a.push_cnt[row], 1);
            if (pos < kSymPush) store_coherent(&a.push_buf[(int64_t)row * kSymPush + pos], (en & 0xffffffff00000000ull) | inv_row);
        }
        const bool own = (lo & kStageOwn) && x > st.thr;      // (the threshold may have risen since the sum was staged)
        const uint64_t mo = __ballot(own);
        if (mo) {
            const int pos = st.cnt + __popcll(mo & ((1ull << lane) - 1ull));
            if (own) cand[pos] = (en & 0xffffffff00000000ull) | (uint32_t)(~row);
            st.cnt += __popcll(mo);
            st.pushed = 1;
            if (st.cnt > kCap - 64) compact<kCap>(cand, st, ntop, lane, false);
        }
    }
    wave_sync();
    fcnt = 0;
}

// One sum of the rare path: staged when it beats the from-row's own threshold or reaches tgu = the (top-byte) minimum
// threshold of the lane's eight slots, as an unsigned number with zeros below the top byte (0xff000000: nothing reaches it).
template <int kCap>
__device__ inline void stage1(uint64_t *cand, TopState &st, uint64_t *fbuf, int &fcnt, int x, uint32_t tgu, int cell, int ntop,
                              int lane, const K3SymArgs &a, uint32_t inv_row)
{
    const bool own = x > st.thr, fgn = (uint32_t)x >= tgu;
    const uint64_t mk = __ballot(own || fgn);
    if (mk) {
        const int pos = fcnt + __popcll(mk & ((1ull << lane) - 1ull));
        if (own || fgn)
            fbuf[pos] = ((uint64_t)(uint32_t)x << 32) | (own ? kStageOwn : 0u) | (fgn ? kStageFgn : 0u) | (uint32_t)cell;
        fcnt += __popcll(mk);
        if (fcnt > kSymF - 64) drain_stage<kCap>(cand, st, fbuf, fcnt, ntop, lane, a, inv_row);
    }
}

// The rare path of a sweep step of pass 1: some lane's maximum beats the own threshold or reaches the lane's minimum.  v0 /
// v1: the lane's eight sums, cell0 / cell1: the cells of v0.x / v1.x; pa / pb / pc: the partial maxima the sweep has anyway (of
// v0.xyz, of v0.w v1.xy, of v1.zw) -- only the sums under a partial maximum that passes are looked at one by one (a step that
// gets here has one such lane, as a rule).
template <int kCap>
__device__ inline void stage8(uint64_t *cand, TopState &st, uint64_t *fbuf, int &fcnt, const int4 &v0, const int4 &v1, int pa,
                              int pb, int pc, uint32_t tgu, int cell0, int cell1, int ntop, int lane, const K3SymArgs &a,
                              uint32_t inv_row)
{
    if (__ballot(pa > st.thr || (uint32_t)pa >= tgu)) {
        stage1<kCap>(cand, st, fbuf, fcnt, v0.x, tgu, cell0, ntop, lane, a, inv_row);
        stage1<kCap>(cand, st, fbuf, fcnt, v0.y, tgu, cell0 + 1, ntop, lane, a, inv_row);
        stage1<kCap>(cand, st, fbuf, fcnt, v0.z, tgu, cell0 + 2, ntop, lane, a, inv_row);
    }
    if (__ballot(pb > st.thr || (uint32_t)pb >= tgu)) {
        stage1<kCap>(cand, st, fbuf, fcnt, v0.w, tgu, cell0 + 3, ntop, lane, a, inv_row);
        stage1<kCap>(cand, st, fbuf, fcnt, v1.x, tgu, cell1, ntop, lane, a, inv_row);
        stage1<kCap>(cand, st, fbuf, fcnt, v1.y, tgu, cell1 + 1, ntop, lane, a, inv_row);
    }
    if (__ballot(pc > st.thr || (uint32_t)pc >= tgu)) {
        stage1<kCap>(cand, st, fbuf, fcnt, v1.z, tgu, cell1 + 2, ntop, lane, a, inv_row);
        stage1<kCap>(cand, st, fbuf, fcnt, v1.w, tgu, cell1 + 3, ntop, lane, a, inv_row);
    }
}

// lanes whose byte 3 of `mx` reaches byte B of `q`: one VOPC instruction with sub-dword operand selection
template <int B> __device__ inline uint64_t top_byte_reaches(int mx, uint32_t q)
{
    uint64_t m;
    if (B == 0) asm("v_cmp_ge_u32_sdwa %0, %1, %2 src0_sel:BYTE_3 src1_sel:BYTE_0" : "=s"(m) : "v"(mx), "v"(q));
    else asm("v_cmp_ge_u32_sdwa %0, %1, %2 src0_sel:BYTE_3 src1_sel:BYTE_1" : "=s"(m) : "v"(mx), "v"(q));
    return m;
}

// sweep_block of k3_core.h for pass 1: the block's accumulators are in threshold order (k3_sym_order), tq = the top bytes of
// gmin of this lane's four sweep steps.  (Two rolled iterations of two steps, like the main kernel's sweep: the code of the
// rare path exists twice, not four times.)
template <int N4, int kCap>
__device__ inline void sweep_block_handover(int4 *acc4, uint64_t *cand, TopState &st, int b, int ntop, int lane, int zero,
                                            uint32_t tq, uint64_t *fbuf, int &fcnt, const K3SymArgs &a, uint32_t inv_row)
{
    static_assert(N4 / 128 == 4, "four sweep steps per block");
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = 2 * h + u;
            const int i0 = t * 128 + lane, i1 = i0 + 64;
            const int4 v0 = acc4[i0], v1 = acc4[i1];
            acc4[i0] = make_int4(zero, zero, zero, zero);
            acc4[i1] = make_int4(zero, zero, zero, zero);
            const int pa = max3i(v0.x, v0.y, v0.z), pb = max3i(v0.w, v1.x, v1.y), pc = max(v1.z, v1.w);
            const int mx = max3i(pa, pb, pc);
            const uint64_t hit = __ballot(mx > st.thr) | (u == 0 ? top_byte_reaches<0>(mx, tq) : top_byte_reaches<1>(mx, tq));
            if (hit)
                stage8<kCap>(cand, st, fbuf, fcnt, v0, v1, pa, pb, pc, (u == 0 ? tq << 24 : tq << 16) & 0xff000000u, b * kSymC + i0 * 4,
                             b * kSymC + i1 * 4, ntop, lane, a, inv_row);
        }
        tq >>= 16;
    }
}

// a row's r-th best: (index, score) of the result -- or, for a part of a job cut over several GPUs, the key itself
__device__ inline void store_result(const K3SymArgs &a, int row, int r, uint64_t key)
{
    if (a.keys_out) {
        a.keys_out[(int64_t)row * a.ntop + r] = key;
    } else {
        const int32_t j = key ? (int32_t)(~(uint32_t)key) : -1;
        const float v = key ? (float)(int32_t)(uint32_t)(key >> 32) * a.inv_scale : 0.f;
        a.out_idx[(int64_t)row * a.ntop + r] = j;
        a.out_val[(int64_t)row * a.ntop + r] = v;
        if (a.host_idx) {          // (visible to the host when the kernel has ended: the word that announces the range is written by a LATER kernel of the stream)
            a.host_idx[(int64_t)row * a.ntop + r] = j;
            a.host_val[(int64_t)row * a.ntop + r] = v;
        }
    }
}

// MODE: the pass (0: own block -> state; 1: the blocks above -> state + pushes; 2: all blocks of the listed rows -> result) --
// a template parameter so that every pass is a kernel of its own name in a trace and carries only its own code
template <int C, int MODE>
__global__ __launch_bounds__(64) void k3_sym_kernel(const K3SymArgs a)
{
    static_assert(C == kSymC, "sym_slot() is written for 2048-row blocks");
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
    if (MODE == 2) {
        // pass 2 is launched blind (the host does not know how many rows the merge listed -- as a rule a handful, often none):
        // a workgroup without an item leaves before it clears 8 KB of accumulators.  (A streamed session launches it once per row
        // range, beside pass 1: twelve times two grids of idle workgroups were 0.5 ms of wave time per match.)
        int n_ovf = a.ovf[0];
        n_ovf = n_ovf > a.n ? a.n : n_ovf;
        const int hi = n_ovf < a.ovf_base + a.ovf_max ? n_ovf : a.ovf_base + a.ovf_max;
        if ((int)blockIdx.x >= (hi > a.ovf_base ? (hi - a.ovf_base) * a.n_sl : 0)) return;
    }
    int4 *acc4 = (int4 *)acc;
    constexpr int N4 = C / 4;
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    for (int t = lane; t < C / 4; t += 64) acc4[t] = make_int4(0, 0, 0, 0);
    wave_sync();
    constexpr int mode = MODE;
    const char *post_bytes = (const char *)(mode == 1 ? a.post_sym : a.post);
    const int src4 = (4 * (lane & 15) + (lane >> 4)) * 4;
    const int sub8 = (lane & 15) * 8;
    const int dummy_addr = 0;          // (piece 0 is the all-zero dummy)
    const int nb = a.nb, ntop = a.ntop;

    int n_items = (a.row_end - a.row_begin + a.n_parts - 1) / a.n_parts + (mode == 1 ? a.n_mag_items : 0);      // (every n_parts-th row)
    const int n_sl = mode == 2 ? a.n_sl : 1;
    const int per_sl = (nb + n_sl - 1) / n_sl;
    if (mode == 2) {
        int n_ovf = a.ovf[0];
        n_ovf = n_ovf > a.n ? a.n : n_ovf;
        const int hi = n_ovf < a.ovf_base + a.ovf_max ? n_ovf : a.ovf_base + a.ovf_max;
        n_items = hi > a.ovf_base ? (hi - a.ovf_base) * n_sl : 0;
    }
    // pass 0, persistent workgroups: a row's dependent loads -- row bounds -> its entries -> their offset-table entries -- are a
    // third of the pass (65 of 200 us with nothing else in it: ~3 us of a wave per row).  The NEXT row's chain is fetched while
    // this row's scatter, sweep and compaction run: the bounds at the top, the entries behind the scatter, the table entries
    // behind the sweep -- each level has arrived when the next one is issued.
    bool pf_ready = false;
    int pf_p0 = 0, pf_p1 = 0, pf_k = 0, pf_cur = 0, pf_nxt = 0;
    float pf_v = 0.f;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        int row = mode == 2 ? a.ovf[1 + a.ovf_base + item / n_sl] : a.row_begin + item * a.n_parts;
        bool magnet = false;      // pass 1: this item is a magnet row x a slice of the blocks BELOW its own
        int m_lo = 0;
        if (mode == 1) {
            if (item < a.n_mag_items) {
                const int ns = (nb - 1 + kSymMagBlocks - 1) / kSymMagBlocks;
                const int mm = item / ns;
                m_lo = (item - mm * ns) * kSymMagBlocks;
                const int mblk = a.mag_b0 + mm / (32 * kSymMag);
                row = __builtin_amdgcn_readfirstlane(a.mag[(int64_t)mblk * (32 * kSymMag) + mm % (32 * kSymMag)]);
                if (row < a.row_begin || row >= a.mag_row_end || m_lo >= row / C || !row_is_mine(a, row)) {     // (-1: no magnet)
                    sym_item_done(a, mblk, lane);
                    continue;
                }
                magnet = true;
            } else {
                row -= a.n_mag_items * a.n_parts;
                if (row >= (nb - 1) * C) continue;      // (the last block's rows have nothing above: not launched by a streamed session)
            }
        }
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
        } else if (mode == 1 && magnet) {
            b_lo = m_lo;
            b_hi = m_lo + kSymMagBlocks < own ? m_lo + kSymMagBlocks : own;
            b_first = b_lo;
        } else if (mode == 1) {
            b_lo = own + 1;
            b_first = b_lo;
        }
        int p0, p1;
        if (mode == 0 && pf_ready) {
            p0 = pf_p0;
            p1 = pf_p1;
        } else {
            p0 = a.a_indptr[row];
            p1 = a.a_indptr[row + 1];
        }
        const bool more = mode == 0 && item + (int)gridDim.x < n_items;
        const int row2 = a.row_begin + (item + (int)gridDim.x) * a.n_parts;      // (pass 0: the row after this one)
        int q0 = 0, q1 = 0;
        if (more) {
            q0 = a.a_indptr[row2];
            q1 = a.a_indptr[row2 + 1];
        }
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
            st.thr = a.thrv[thr_pos(a, row)];
            // (a magnet item starts from the threshold alone: what it finds goes to the row's push slots, the keys stay with the row's own item)
            const uint64_t k = lane < ntop && !magnet ? a.keys[(int64_t)row * ntop + lane] : 0ull;
            st.cnt = __popcll(__ballot(k != 0ull));
            if (k) cand[lane] = k;
            warmed = true;
            wave_sync();
        }
        int fcnt = 0;

        const int n_blk = b_hi - b_lo;
        int cur0 = 0, nxt0 = 0;
        float as0 = 0.f;
        const bool have0 = lane < nnz;
        const int32_t *trow = a.tab;
        if (have0 && mode == 0 && pf_ready) {
            as0 = pf_v * a.scale;
            trow = a.tab + (int64_t)pf_k * nb;
            cur0 = pf_cur;
            nxt0 = pf_nxt;
        } else if (have0) {
            as0 = a.a_val[p0 + lane] * a.scale;
            trow = a.tab + (int64_t)a.a_idx[p0 + lane] * nb;
            cur0 = trow[b_first];
            nxt0 = trow[b_first + 1];
        }
        const bool have2 = more && lane < q1 - q0;

        uint32_t tq_next = 0xffffffffu;      // (0xff: no sum reaches it)
#if PFZ_K3_SYM_EXP != 1
        if (mode == 1 && !magnet) tq_next = a.gmin[(int64_t)b_first * 64 + lane];
#endif
        for (int it = 0, b = b_first; it < n_blk; ++it) {
            const int s = cur0, e = have0 ? nxt0 : cur0;
            const int b_next = b + 1 < b_hi ? b + 1 : b_lo;
            // pass 1: the minimum thresholds of this lane's four sweep steps were fetched one block ahead, like the table entries:
            // every round of the scatter begins with s_waitcnt vmcnt(0) (the compiler's), so a load issued just before the
            // scatter has its whole latency exposed there (measured: a tenth of pass 1)
            const uint32_t tq = tq_next;
            bool touched = __ballot(e > s) != 0;
#if PFZ_K3_SYM_EXP == 13     // (what-if 13: pass 0 without its scatter)
            if (touched && mode != 0) scatter_pieces(acc, post_bytes, mark, e - s, s, as0, lane, src4, sub8, dummy_addr);
#else
            if (touched) scatter_pieces(acc, post_bytes, mark, e - s, s, as0, lane, src4, sub8, dummy_addr);
#endif
            if (have0 && it + 1 < n_blk) {
                cur0 = trow[b_next];
                nxt0 = trow[b_next + 1];
            }
            if (mode == 0 && have2) {           // (level 2 of the next row's chain: its bounds are here by now)
                pf_k = a.a_idx[q0 + lane];
                pf_v = a.a_val[q0 + lane];
            }
#if PFZ_K3_SYM_EXP != 1      // (what-if 1: nothing is handed over)
            if (mode == 1 && !magnet && it + 1 < n_blk) tq_next = a.gmin[(int64_t)b_next * 64 + lane];
#endif
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
#if PFZ_K3_SYM_EXP != 10     // (what-if 10: no warm start)
                    if (ntop <= kWarmMaxTop) {
                        const int t = warm_threshold<N4>(acc4, 0, ntop + 1, lane);
                        st.thr = t > st.thr ? t : st.thr;
                    }
#endif
                }
                if (mode == 1)
                    sweep_block_handover<N4, kSymCap>(acc4, cand, st, b, ntop, lane, zero, tq, fbuf, fcnt, a, inv_row);
#if PFZ_K3_SYM_EXP == 12     // (what-if 12: pass 0 without its sweep)
                else if (mode != 0)
#else
                else
#endif
                    sweep_block<N4, kSymCap>(acc4, cand, st, b * C, self_col, ntop, lane, zero);
                wave_sync();
            }
            if (mode == 0 && have2) {           // (level 3: behind the sweep, ahead of the compaction and the stores)
                const int32_t *t2 = a.tab + (int64_t)pf_k * nb + row2 / C;
                pf_cur = t2[0];
                pf_nxt = t2[1];
            }
            b = b_next;
        }
        if (mode == 0) {
            pf_ready = more;
            pf_p0 = q0;
            pf_p1 = q1;
        }

        if (mode == 1 && fcnt) drain_stage<kSymCap>(cand, st, fbuf, fcnt, ntop, lane, a, inv_row);
#if PFZ_K3_SYM_EXP == 11     // (what-if 11: pass 0 without its final compaction)
        if (mode != 0)
#endif
        compact<kSymCap>(cand, st, ntop, lane);
        if (mode == 1 && magnet) {
            // what the row found below its own block, to its own push slots (the merge joins them with the row's keys)
            if (lane < st.cnt) {
                const int pos = atomicAdd(&a.push_cnt[row], 1);
                if (pos < kSymPush) store_coherent(&a.push_buf[(int64_t)row * kSymPush + pos], cand[lane]);
            }
        } else if (mode == 2 && n_sl > 1) {
            if (lane < ntop) a.part[(int64_t)item * ntop + lane] = lane < st.cnt ? cand[lane] : 0ull;
        } else if (mode == 2) {
            for (int r = lane; r < ntop; r += 64) store_result(a, row, r, r < st.cnt ? cand[r] : 0ull);
        } else {
            if (lane < ntop) {
                const uint64_t k = lane < st.cnt ? cand[lane] : 0ull;
                if (mode == 1) store_coherent(&a.keys[(int64_t)row * ntop + lane], k);
                else a.keys[(int64_t)row * ntop + lane] = k;
            }
            if (mode == 0 && lane == 0) a.thrv[thr_pos(a, row)] = st.thr;
        }
        if (mode == 1) sym_item_done(a, own, lane);
        wave_sync();    // cand is reused by the next item
    }
}

// own keys + pushed keys -> the row's sorted top-n (one wave per row); rows that were sent more than kSymPush are listed
// (kWaves rows per workgroup.  4 as a rule; 1 for a streamed session, whose merges run BESIDE pass 1: a CU's LDS is full of pass-1
// workgroups of ~10 KB each, and only a workgroup that fits the hole ONE of them leaves -- 4.6 KB here, not 18 -- gets in before
// pass 1 has finished.  Measured: the four-row merge of the first range sat in its queue for 1.2 ms, stream priority or not.)
#ifndef PFZ_K3_SYM_MERGE_W
#define PFZ_K3_SYM_MERGE_W 1       // rows per workgroup of a streamed range's merge (tuning builds: 2 still fits the hole, 9.2 KB)
#endif
template <int kWaves>
__global__ __launch_bounds__(64 * kWaves) void k3_sym_merge(const K3SymArgs a)
{
    __shared__ __attribute__((aligned(16))) uint64_t cand_all[kWaves][kSymMergeCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = a.row_begin + blockIdx.x * kWaves + wave;
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
    // (a part of a job cut over several GPUs: the own-block keys of a row count on the part the row belongs to)
    const uint64_t k = lane < ntop && row_is_mine(a, row) ? a.keys[(int64_t)row * ntop + lane] : 0ull;
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
    for (int r = lane; r < ntop; r += 64) store_result(a, row, r, r < st.cnt ? cand[r] : 0ull);
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
        for (int q = lane; q < ntop; q += 64) store_result(a, row, q, q < st.cnt ? cand[q] : 0ull);
        wave_sync();
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------

struct K3SymState {
    pfz_ctx *ctx = nullptr;
    int64_t n = 0;
    int32_t *thrv = nullptr;
    uint16_t *slot4 = nullptr;
    uint32_t *gmin = nullptr;
    int32_t *thr_slot = nullptr;
    uint16_t *row_slot = nullptr;
    int32_t *mag = nullptr;
    int2 *post_sym = nullptr;
    uint64_t *keys = nullptr;
    int32_t *push_cnt = nullptr;
    uint64_t *push_buf = nullptr;
    int32_t *ovf = nullptr;
    uint64_t *part = nullptr;
    int32_t *ovfx[3] = {nullptr, nullptr, nullptr};      // streamed sessions: the further side streams' lists of rows to recompute and partial lists
    uint64_t *partx[3] = {nullptr, nullptr, nullptr};
    uint32_t *done = nullptr;            // [nb] streamed sessions: pass-1 items finished per block
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
    void *const bufs[] = {s->thrv, s->slot4, s->gmin, s->thr_slot, s->row_slot, s->mag, s->post_sym, s->keys, s->push_cnt, s->push_buf, s->ovf, s->part, s->ovfx[0], s->partx[0], s->ovfx[1], s->partx[1], s->ovfx[2], s->partx[2], s->done};
    for (void *p : bufs)
        if (p) pool_free(p);
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
    if (ix->block_cols != kSymC || !ix->pblk || ntop > kSymKeep || ix->n_blocks < 2 || ix->n_rows >= ((int64_t)1 << 30)) return 0;
    const K3SymState *s = ix->sym;
    if (s && s->n < 0) return 0;       // (the session buffers could not be allocated once: the row-major kernel serves this index)
    if (row_begin > 0) {
        const bool cont = s && s->next_row == row_begin && s->a_serial == A->serial && s->out == out && s->ntop == ntop &&
                          s->thr0 == thr0 && s->scale == scale;
        return cont ? 2 : 0;
    }
    if (force == 1) return 1;
    // auto: where halving K3 pays for seven more launches and the state round trip, and where the row-major kernel is the one that
    // would run (k3_lockstep.hip takes the to-sides beyond 250 000 rows); a first range of less than a fifth of the rows is a
    // shard of a bigger job (bench --scaling strong), not the start of a whole self-match
    if (ix->n_rows < sym_env_int("PFZ_K3_SYM_MIN", 20480) || ix->n_rows > 250000) return 0;
    return (row_end - row_begin) * 5 >= ix->n_rows ? 1 : 0;
}

// the session buffers of an index, allocated by its first symmetric launch and kept with it
static int sym_state_alloc(pfz_ctx *ctx, const pfz_index *ix, K3SymState *s)
{
    const int64_t n = ix->n_rows;
    const size_t cells = (size_t)ix->n_blocks * kSymC;
    PFZ_TRY(pool_alloc(ctx, &s->thrv, (size_t)(n + 64) * sizeof(int32_t)));      // (part by part, every part's stretch rounded up: k3_sym_sharded)
    PFZ_TRY(pool_alloc(ctx, &s->slot4, cells * sizeof(uint16_t)));
    PFZ_TRY(pool_alloc(ctx, &s->gmin, cells / 32 * sizeof(uint32_t)));
    PFZ_TRY(pool_alloc(ctx, &s->thr_slot, cells * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &s->row_slot, cells * sizeof(uint16_t)));
    PFZ_TRY(pool_alloc(ctx, &s->mag, (size_t)ix->n_blocks * 32 * kSymMag * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &s->post_sym, (size_t)ix->piece_cap * kPiece * sizeof(int2)));
    PFZ_TRY(pool_alloc(ctx, &s->keys, (size_t)n * kSymKeep * sizeof(uint64_t)));
    PFZ_TRY(pool_alloc(ctx, &s->push_cnt, (size_t)n * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &s->push_buf, (size_t)n * kSymPush * sizeof(uint64_t)));
    PFZ_TRY(pool_alloc(ctx, &s->ovf, (size_t)(n + 1) * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &s->part, (size_t)kSymSlicedRows * kSymSlices * kSymKeep * sizeof(uint64_t)));
    for (int q = 0; q + 1 < kSymSides; ++q) {
        PFZ_TRY(pool_alloc(ctx, &s->ovfx[q], (size_t)(n + 1) * sizeof(int32_t)));
        PFZ_TRY(pool_alloc(ctx, &s->partx[q], (size_t)kSymSlicedRows * kSymSlices * kSymKeep * sizeof(uint64_t)));
    }
    PFZ_TRY(pool_alloc(ctx, &s->done, (size_t)(ix->n_blocks + 64) * kSymDoneStride * sizeof(uint32_t)));
    return PFZ_OK;
}

// the session state of an index, created on first use; NULL: its buffers could not be allocated (now or earlier) -- nothing was
// enqueued, the row-major kernel serves this index
static K3SymState *sym_state_of(pfz_ctx *ctx, const pfz_index *ix)
{
    K3SymState *s = ix->sym;
    if (s) return s->n < 0 ? nullptr : s;
    s = new K3SymState();
    s->ctx = ctx;
    s->n = ix->n_rows;
    ix->sym = s;      // (freed with the index, whatever happens below)
    if (getenv("PFZ_K3_SYM_FAIL_ALLOC") || sym_state_alloc(ctx, ix, s) != PFZ_OK) {      // (the knob: tests of this fallback)
        void *const bufs[] = {s->thrv, s->slot4, s->gmin, s->thr_slot, s->row_slot, s->mag, s->post_sym, s->keys, s->push_cnt, s->push_buf, s->ovf, s->part, s->ovfx[0], s->partx[0], s->ovfx[1], s->partx[1], s->ovfx[2], s->partx[2], s->done};
        for (void *p : bufs)
            if (p) pool_free(p);
        *s = K3SymState();
        s->ctx = ctx;
        s->n = -1;
        (void)hipGetLastError();
        return nullptr;
    }
    return s;
}

// pairs of postings k3_sym_repost has to look at: the index' pieces if the host knows them by now (the build enqueued their count
// a pass 0 ago: asking does not wait), its piece capacity otherwise (the kernel checks against the device's count)
static int64_t sym_repost_pairs(const pfz_index *ix)
{
    if (ix->pieces_lazy.pending && hipEventQuery(ix->pieces_lazy.ev) == hipSuccess) (void)index_ready(ix);
    (void)hipGetLastError();      // (hipErrorNotReady is not an error)
    return (ix->pieces_lazy.pending ? ix->piece_cap : (int64_t)ix->n_pieces + 1) * (kPiece / 2);
}

// the arguments every launch of a job shares
static void sym_fill_args(K3SymArgs &a, const pfz_index *ix, const pfz_csr *A, K3SymState *s, int32_t ntop, int32_t thr0, float scale,
                          float inv_scale)
{
    a.a_indptr = A->indptr;
    a.a_idx = A->indices;
    a.a_val = A->data;
    a.n = (int32_t)ix->n_rows;
    a.tab = ix->tab;
    a.post = ix->post;
    a.post_sym = s->post_sym;
    a.pblk = ix->pblk;
    a.nb = ix->n_blocks;
    a.n_pieces1 = ix->tab + ix->n_cols * ix->n_blocks;
    a.ntop = ntop;
    a.thr0 = thr0;
    a.scale = scale;
    a.inv_scale = inv_scale;
    a.row_begin = 0;
    a.row_end = (int32_t)ix->n_rows;
    a.n_parts = 1;
    a.my_part = 0;
    a.per = (int32_t)ix->n_rows;
    a.thr_by_part = 0;
    a.keys_out = nullptr;
    a.thrv = s->thrv;
    a.slot4 = s->slot4;
    a.gmin = s->gmin;
    a.thr_slot = s->thr_slot;
    a.row_slot = s->row_slot;
    a.mag = s->mag;
    a.n_mag_items = 0;
    a.mag_b0 = 0;
    a.mag_row_end = 0;
    a.keys = s->keys;
    a.push_cnt = s->push_cnt;
    a.push_buf = s->push_buf;
    a.ovf = s->ovf;
    a.part = s->part;
    a.ovf_base = 0;
    a.ovf_max = 0;
    a.n_sl = 1;
    a.out_idx = nullptr;
    a.out_val = nullptr;
    a.done = nullptr;
    a.blk_lo = a.blk_hi = 0;
    a.host_idx = nullptr;
    a.host_val = nullptr;
}

// pass 2 of the rows the merge listed (sent more than their push slots hold): the first kSymSlicedRows in slices of the to-blocks
// (a whole row is ~100 us of one wave: a handful of rows would cost that much wall time), their partial lists merged; whatever is
// listed beyond, as whole rows
static void sym_launch_pass2(pfz_ctx *ctx, K3SymArgs a, hipStream_t stream = nullptr, int per_cu = 16)
{
    if (!stream) stream = ctx->stream;
    const int nb = a.nb;
    const unsigned grid2 = (unsigned)ctx->prop.multiProcessorCount * (unsigned)per_cu;
    const int per = (nb + kSymSlices - 1) / kSymSlices;
    a.n_mag_items = 0;
    a.n_parts = 1;         // (the listed rows are recomputed in full, whoever they belong to)
    a.n_sl = (nb + per - 1) / per;
    a.ovf_base = 0;
    a.ovf_max = kSymSlicedRows;
    if (a.n_sl > 1) {
        hipLaunchKernelGGL((k3_sym_kernel<kSymC, 2>), dim3(grid2), dim3(64), 0, stream, a);
        hipLaunchKernelGGL(k3_sym_merge_slices, dim3(256), dim3(256), 0, stream, a);
        a.ovf_base = kSymSlicedRows;
    }
    a.n_sl = 1;
    a.ovf_max = a.n;
    hipLaunchKernelGGL((k3_sym_kernel<kSymC, 2>), dim3(grid2), dim3(64), 0, stream, a);
}

// *declined: the buffers of the session could not be allocated -- nothing was enqueued, the caller runs the row-major kernel
// (which needs none of them) and this index is not asked again
int k3_sym_launch(pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int64_t row_begin, int64_t row_end, int32_t ntop,
                  int32_t thr0, float scale, float inv_scale, pfz_topn *out, bool start, bool *declined)
{
    const int64_t n = ix->n_rows;
    const int nb = ix->n_blocks;
    *declined = false;
    K3SymState *s = sym_state_of(ctx, ix);
    if (!s) {
        *declined = true;
        return PFZ_OK;
    }
    s->next_row = -1;     // (no session while this call can still fail)
    K3SymArgs a;
    sym_fill_args(a, ix, A, s, ntop, thr0, scale, inv_scale);
    a.out_idx = out->idx;
    a.out_val = out->val;
    if (start) {
        // pass 0 over ALL rows: every row's first threshold is there before anybody hands anything over; then the re-deal of
        // every block's rows to the accumulator slots and the re-dealt copy of the postings (k3_sym_order also clears push_cnt)
        a.row_begin = 0;
        a.row_end = (int32_t)n;
        // (workgroups that loop over a few rows each: the loop is what lets a row's loads be fetched under the row before it.
        // 128 per CU, ~3 rows each at 100 000 rows: exactly the 18 that are resident -- 22 rows each, every load prefetched --
        // was SLOWER, K3 1.842 ms against 1.771, the rows' costs differ too much for a fixed deal; 36 / 72 / 144 per CU
        // 1.780 / 1.758 / 1.757; a workgroup per row, as before, 1.768 - 1.771)
        const unsigned grid0 = (unsigned)std::min<int64_t>(n, (int64_t)ctx->prop.multiProcessorCount * kSymP0PerCu);
        hipLaunchKernelGGL((k3_sym_kernel<kSymC, 0>), dim3(grid0), dim3(64), 0, ctx->stream, a);
        hipLaunchKernelGGL(k3_sym_order, dim3((unsigned)nb), dim3(1024), 0, ctx->stream, a);
        const int64_t pairs = sym_repost_pairs(ix);
        hipLaunchKernelGGL(k3_sym_repost, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, ctx->stream, a);
    }
    PFZ_HIP(hipMemsetAsync(s->ovf, 0, sizeof(int32_t), ctx->stream));
    // pass 1: the rows of this range that have blocks above their own
    const int64_t last_block_row = (int64_t)(nb - 1) * kSymC;
    a.row_begin = (int32_t)row_begin;
    a.row_end = (int32_t)(row_end < last_block_row ? row_end : last_block_row);
    if (a.row_end < a.row_begin) a.row_end = a.row_begin;
    // ... and, first in the grid (they are the long ones), the magnet rows of the range x slices of the blocks below their own
    a.mag_b0 = (int32_t)(row_begin / kSymC);
    a.mag_row_end = (int32_t)row_end;
    a.n_mag_items = (int32_t)(((row_end - 1) / kSymC - a.mag_b0 + 1) * 32 * kSymMag * ((nb - 1 + kSymMagBlocks - 1) / kSymMagBlocks));
#ifdef PFZ_EXPERIMENTS
    // tools/predict_scaling.py: pass 1 of ONE part of a job cut over N GPUs, alone on this GPU ("p/N"; whole jobs only; the result is
    // that part's share -- wrong on purpose, compiled into variant builds only)
    if (const char *solo = getenv("PFZ_K3_SYM_SOLO")) {
        int p = 0, np_ = 1;
        if (sscanf(solo, "%d/%d", &p, &np_) == 2 && np_ > 1 && p >= 0 && p < np_ && row_begin == 0) {
            a.n_parts = np_;
            a.my_part = p;
            a.row_begin = p;
        }
    }
    const int rows1 = a.row_end > a.row_begin ? (a.row_end - a.row_begin + a.n_parts - 1) / a.n_parts : 0;
    if (rows1 + a.n_mag_items > 0)
        hipLaunchKernelGGL((k3_sym_kernel<kSymC, 1>), dim3((unsigned)(rows1 + a.n_mag_items)), dim3(64), 0, ctx->stream, a);
    a.n_parts = 1;
    a.my_part = 0;
    if (false)
#endif
    if (a.row_end - a.row_begin + a.n_mag_items > 0)      // (one row per one-wave workgroup: 2 / 4 / 22 rows per workgroup measured no faster -- the dispatcher is not what a row waits for)
        hipLaunchKernelGGL((k3_sym_kernel<kSymC, 1>), dim3((unsigned)(a.row_end - a.row_begin + a.n_mag_items)), dim3(64), 0, ctx->stream, a);
    a.n_mag_items = 0;
    // merge, then the rows that were sent too much
    a.row_begin = (int32_t)row_begin;
    a.row_end = (int32_t)row_end;
    hipLaunchKernelGGL(k3_sym_merge<4>, dim3((unsigned)((row_end - row_begin + 3) / 4)), dim3(256), 0, ctx->stream, a);
    sym_launch_pass2(ctx, a);
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

// ---- the whole job in ONE pass-1 launch, its row ranges handed on as they finish --------------------------------------------------
// What TFIDF.match wants from a big self-match is its rows in ascending ranges, each as soon as it is final, so that the frame's
// columns are built while the device works on.  Round 5 enqueued one session launch per range: every range paid its own pass-1
// tail (a launch ends on its slowest rows), merge and overflow pass -- 2.2 ms of K3 in four ranges where one launch takes 1.8.
// Here pass 1 is ONE launch over all rows (and all magnet items) on the context's stream; every item counts itself on `done[block
// of its row]`; on a side stream, per range: k3_sym_wait (one wave) until the blocks up to the range's end are complete -- whoever
// pushes to a row sits in a lower or the same block --, the range's merge, its overflow pass, and the event the host's download
// waits for.  ends[i]: multiples of 2048 except the last (= n).  Results: the session's, bit for bit.
int k3_sym_launch_streamed(pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int32_t ntop, int32_t thr0, float scale, float inv_scale,
                           pfz_topn *out, int32_t n_ranges, const int64_t *ends, int32_t first_event, int32_t *host_idx, float *host_val,
                           bool *declined)
{
    const int64_t n = ix->n_rows;
    const int nb = ix->n_blocks;
    *declined = false;
    K3SymState *s = sym_state_of(ctx, ix);
    if (!s) {
        *declined = true;
        return PFZ_OK;
    }
    if (!ctx->stream3) {
        // HIGH priority: the merges are a few microseconds of work that must get wave slots while pass 1 still has tens of thousands of
        // workgroups to dispatch (at equal priority the side stream's kernels were served when pass 1 had finished: measured)
        int pr_lo = 0, pr_hi = 0;
        PFZ_HIP(hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi));
        PFZ_HIP(hipStreamCreateWithPriority(&ctx->stream3, hipStreamNonBlocking, pr_hi));
        PFZ_HIP(hipEventCreateWithFlags(&ctx->ev3, hipEventDisableTiming));
        for (int q = 0; q + 1 < kSymSides; ++q) {
            PFZ_HIP(hipStreamCreateWithPriority(&ctx->stream3x[q], hipStreamNonBlocking, pr_hi));
            PFZ_HIP(hipEventCreateWithFlags(&ctx->ev3x[q], hipEventDisableTiming));
        }
    }
    s->next_row = -1;
    K3SymArgs a;
    sym_fill_args(a, ix, A, s, ntop, thr0, scale, inv_scale);
    a.out_idx = out->idx;
    a.out_val = out->val;
    a.host_idx = host_idx;
    a.host_val = host_val;
    a.done = s->done;          // (k3_sym_order clears push_cnt and the blocks' counters: no memset of their own in front of pass 0)
    const unsigned grid0 = (unsigned)std::min<int64_t>(n, (int64_t)ctx->prop.multiProcessorCount * kSymP0PerCu);
    hipLaunchKernelGGL((k3_sym_kernel<kSymC, 0>), dim3(grid0), dim3(64), 0, ctx->stream, a);
    hipLaunchKernelGGL(k3_sym_order, dim3((unsigned)nb), dim3(1024), 0, ctx->stream, a);
    const int64_t pairs = sym_repost_pairs(ix);
    hipLaunchKernelGGL(k3_sym_repost, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, ctx->stream, a);
    PFZ_HIP(hipEventRecord(ctx->ev3, ctx->stream));            // (the counters are zero, the slots re-dealt)
    // pass 1: every row that has blocks above + the magnet items of every block, one launch
    const int64_t last_block_row = (int64_t)(nb - 1) * kSymC;
    a.row_begin = 0;
    a.row_end = (int32_t)last_block_row;
    a.mag_b0 = 0;
    a.mag_row_end = (int32_t)n;
    a.n_mag_items = (int32_t)((int64_t)nb * 32 * kSymMag * ((nb - 1 + kSymMagBlocks - 1) / kSymMagBlocks));
    a.done = s->done;
    hipLaunchKernelGGL((k3_sym_kernel<kSymC, 1>), dim3((unsigned)(a.row_end + a.n_mag_items)), dim3(64), 0, ctx->stream, a);
    PFZ_HIP(hipGetLastError());
    a.n_mag_items = 0;
    // The ranges, on kSymSides side streams in turn.  A range's chain -- wait, merge, overflow pass in slices, their merge, the overflow
    // pass of whole rows -- is ~0.1 ms of small kernels one behind the other, and the last ranges of a list become final within
    // less than that of each other (a row walks the blocks from its own upwards: 1 - (1 - x)^2 of pass 1 is spent when a share x of
    // the rows is done): on one stream the chains of the last three ranges queued up behind pass 1's end, 0.2 ms; going round
    // several streams a chain runs beside its neighbours'.  Each stream has its own list of rows to recompute and its own partial
    // lists; a range is announced by the wait kernel of the NEXT range on ITS stream (the stream is in order); a range's wait
    // covers the blocks from the end of the range before it ON ITS STREAM (the ranges between wait on the other streams).
    hipStream_t sides[4] = {ctx->stream3, ctx->stream3x[0], ctx->stream3x[1], ctx->stream3x[2]};
    hipEvent_t side_ev[4] = {ctx->ev3, ctx->ev3x[0], ctx->ev3x[1], ctx->ev3x[2]};
    const int n_sides = n_ranges < kSymSides ? n_ranges : kSymSides;
    for (int q = 0; q < n_sides; ++q) PFZ_HIP(hipStreamWaitEvent(sides[q], ctx->ev3, 0));
    int64_t row0 = 0;
    int64_t covered[4] = {0, 0, 0, 0};        // per stream: the row its waits have covered the blocks up to
    int32_t *flag[4] = {nullptr, nullptr, nullptr, nullptr};
    int32_t flag_value[4] = {0, 0, 0, 0};
    for (int32_t i = 0; i < n_ranges; ++i) {
        const int q = i % n_sides;
        hipStream_t side = sides[q];
        const int64_t row1 = ends[i];
        a.ovf = q ? s->ovfx[q - 1] : s->ovf;
        a.part = q ? s->partx[q - 1] : s->part;
        a.blk_lo = (int32_t)(covered[q] / kSymC);
        a.blk_hi = (int32_t)((row1 + kSymC - 1) / kSymC);
        covered[q] = row1;
        hipLaunchKernelGGL(k3_sym_wait, dim3(1), dim3(64), 0, side, a, flag[q], flag_value[q]);
        a.row_begin = (int32_t)row0;
        a.row_end = (int32_t)row1;
        hipLaunchKernelGGL(k3_sym_merge<PFZ_K3_SYM_MERGE_W>, dim3((unsigned)((row1 - row0 + PFZ_K3_SYM_MERGE_W - 1) / PFZ_K3_SYM_MERGE_W)),
                           dim3(64 * PFZ_K3_SYM_MERGE_W), 0, side, a);
        sym_launch_pass2(ctx, a, side, 2);       // (a range's overflow rows are few: two workgroups per CU loop over them)
        PFZ_HIP(hipGetLastError());
        PFZ_HIP(hipEventRecord(ctx->events[first_event + i], side));
        PFZ_TRY(event_flag_next(ctx, first_event + i, &flag[q], &flag_value[q]));       // (... and as a word in pinned memory: pfz_topn_rows_begin / _finish)
        row0 = row1;
    }
    a.blk_lo = a.blk_hi = 0;
    a.ovf = s->ovf;
    a.part = s->part;
    for (int q = 0; q < n_sides; ++q) {
        hipLaunchKernelGGL(k3_sym_wait, dim3(1), dim3(64), 0, sides[q], a, flag[q], flag_value[q]);      // (the stream's last range's word)
        PFZ_HIP(hipGetLastError());
        PFZ_HIP(hipEventRecord(side_ev[q], sides[q]));
    }
    // whatever follows on the context's stream (and its timers) comes after the last range
    for (int q = 0; q < n_sides; ++q) PFZ_HIP(hipStreamWaitEvent(ctx->stream, side_ev[q], 0));      // (ev3: free again once pass 1 had been launched behind it)
    s->next_row = n;
    s->launches += 1;
    s->rows += n;
    s->a_serial = A->serial;
    s->out = out;
    s->ntop = ntop;
    s->thr0 = thr0;
    s->scale = scale;
    return PFZ_OK;
}

// ---- the self-match cut over several GPUs ----------------------------------------------------------------
// Every GPU holds the whole list's matrix and index (replicated).  Part p of n_parts works on the rows p, p + n_parts, ...: their
// pass 0 -- the thresholds of the parts are all-gathered, the re-deal of the slots is the same everywhere --, their pass 1 and
// their magnet items; every unordered pair of rows is still scored once over all parts.  What a part finds for a row -- its
// own rows' keys, what its rows handed to ANY row -- it merges into one sorted list of at most ntop keys per row; the parts'
// lists are all-gathered (n x ntop x 8 bytes per part: 4 MB at 100 000 x 5) and every GPU merges them: the full result on
// every GPU, bit-identical to one GPU's (the selection is by exact key -- integer sum, column -- on every level; a row that one
// part recomputes in full brings duplicates of the others' keys, which a selection round removes together).
__global__ __launch_bounds__(256) void k3_sym_merge_parts(const uint64_t *__restrict__ keys_all, int32_t n_parts, int32_t n,
                                                           int32_t ntop, float inv_scale, int32_t *__restrict__ out_idx,
                                                           float *__restrict__ out_val)
{
    __shared__ __attribute__((aligned(16))) uint64_t cand_all[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n) return;
    uint64_t *cand = cand_all[wave];
    TopState st;
    st.cnt = 0;
    st.thr = 0;
    st.pushed = 0;
    const int total = n_parts * ntop;       // <= 256 (k3_sym_sharded_ok)
    for (int e0 = 0; e0 < total; e0 += 64) {
        const int e = e0 + lane;
        const uint64_t k = e < total ? keys_all[((int64_t)(e / ntop) * n + row) * ntop + e % ntop] : 0ull;
        const uint64_t mk = __ballot(k != 0ull);
        if (k) cand[st.cnt + __popcll(mk & ((1ull << lane) - 1ull))] = k;
        st.cnt += __popcll(mk);
    }
    compact<256>(cand, st, ntop, lane);
    for (int q = lane; q < ntop; q += 64) {
        const uint64_t key = q < st.cnt ? cand[q] : 0ull;
        out_idx[(int64_t)row * ntop + q] = key ? (int32_t)(~(uint32_t)key) : -1;
        out_val[(int64_t)row * ntop + q] = key ? (float)(int32_t)(uint32_t)(key >> 32) * inv_scale : 0.f;
    }
}

bool k3_sym_sharded_ok(const pfz_index *ix, const pfz_csr *A, int32_t ntop, int32_t n_parts)
{
    if (A->serial != ix->src_serial || A->n_rows != ix->n_rows) return false;
    if (ix->block_cols != kSymC || !ix->pblk || ntop > kSymKeep || ix->n_blocks < 2 || ix->n_rows >= ((int64_t)1 << 30)) return false;
    if (n_parts < 1 || (int64_t)n_parts * ntop > 256 || n_parts > 64) return false;
    if (ix->sym && ix->sym->n < 0) return false;
    const int force = sym_env_int("PFZ_K3_SYM", -1);      // 0: never; 1: whenever the arithmetic allows (tests); default: by size, as on one GPU
    if (force >= 0) return force != 0;
    return ix->n_rows >= sym_env_int("PFZ_K3_SYM_MIN", 20480) && ix->n_rows <= 250000;
}

// Every allocation of the job happens BEFORE its first collective, and a rank that fails there (or in any launch behind) tears the
// communicator down (comm_abort): its peers' collectives end in an RCCL error instead of waiting for a rank that has left.  That
// the ranks agree on taking this form at all -- environment, sizes, whether the session buffers exist -- is settled by
// pfz_comm_symmetric_ok, once per job and before the first call.
static int k3_sym_sharded_body(pfz_ctx *ctx, pfz_comm *comm, const pfz_index *ix, const pfz_csr *A, int32_t ntop, float lower_bound,
                               pfz_topn *out);

int k3_sym_sharded(pfz_ctx *ctx, pfz_comm *comm, const pfz_index *ix, const pfz_csr *A, int32_t ntop, float lower_bound, pfz_topn *out)
{
    const int rc = k3_sym_sharded_body(ctx, comm, ix, A, ntop, lower_bound, out);
    if (rc != PFZ_OK && comm_world(comm) > 1) comm_abort(comm);
    return rc;
}

static int k3_sym_sharded_body(pfz_ctx *ctx, pfz_comm *comm, const pfz_index *ix, const pfz_csr *A, int32_t ntop, float lower_bound,
                               pfz_topn *out)
{
    const int64_t n = ix->n_rows;
    const int nb = ix->n_blocks;
    const int n_parts = comm_world(comm), part = comm_rank(comm);
    K3SymState *s = sym_state_of(ctx, ix);
    if (!s) {
        set_error("pfz_comm_cossim_topn_symmetric: the session buffers of this index could not be allocated");
        return PFZ_ERR_NOMEM;
    }
    float scale, inv_scale;
    int32_t thr0;
    k3_fixed_point(A, ix, lower_bound, &scale, &inv_scale, &thr0);
    s->next_row = -1;
    struct Tmp {
        uint64_t *p = nullptr;
        ~Tmp() { if (p) pool_free(p); }
    } mine, all;
    const size_t list_bytes = (size_t)n * ntop * sizeof(uint64_t);
    PFZ_TRY(pool_alloc(ctx, &mine.p, list_bytes));
    PFZ_TRY(pool_alloc(ctx, &all.p, list_bytes * n_parts));
    K3SymArgs a;
    sym_fill_args(a, ix, A, s, ntop, thr0, scale, inv_scale);
    a.n_parts = n_parts;
    a.my_part = part;
    a.per = (int32_t)((n + n_parts - 1) / n_parts);
    a.thr_by_part = 1;
    a.keys_out = mine.p;
    PFZ_HIP(hipMemsetAsync(s->ovf, 0, sizeof(int32_t), ctx->stream));      // (push_cnt: cleared by k3_sym_order)
    // pass 0 of this part's rows; the parts' thresholds, stretch by stretch, complete thrv on every GPU
    a.row_begin = part;
    a.row_end = (int32_t)n;
    const unsigned mine0 = (unsigned)((n - part + n_parts - 1) / n_parts);
    const unsigned grid0 = std::min<unsigned>(mine0, (unsigned)ctx->prop.multiProcessorCount * (unsigned)kSymP0PerCu);
    if (mine0) hipLaunchKernelGGL((k3_sym_kernel<kSymC, 0>), dim3(grid0), dim3(64), 0, ctx->stream, a);
    PFZ_HIP(hipGetLastError());
    if (n_parts > 1) PFZ_TRY(comm_allgather_bytes(comm, s->thrv + (size_t)part * a.per, s->thrv, (size_t)a.per * sizeof(int32_t)));
    hipLaunchKernelGGL(k3_sym_order, dim3((unsigned)nb), dim3(1024), 0, ctx->stream, a);
    const int64_t pairs = sym_repost_pairs(ix);
    hipLaunchKernelGGL(k3_sym_repost, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, ctx->stream, a);
    // pass 1 of this part's rows and magnets
    const int64_t last_block_row = (int64_t)(nb - 1) * kSymC;
    a.row_end = (int32_t)(n < last_block_row ? n : last_block_row);
    const int rows1 = a.row_end > part ? (a.row_end - part + n_parts - 1) / n_parts : 0;
    if (a.row_end < a.row_begin) a.row_end = a.row_begin;
    a.mag_b0 = 0;
    a.mag_row_end = (int32_t)n;
    a.n_mag_items = (int32_t)(nb * 32 * kSymMag * ((nb - 1 + kSymMagBlocks - 1) / kSymMagBlocks));
    hipLaunchKernelGGL((k3_sym_kernel<kSymC, 1>), dim3((unsigned)(rows1 + a.n_mag_items)), dim3(64), 0, ctx->stream, a);
    a.n_mag_items = 0;
    // this part's list of every row; the rows it was sent too much for, in full
    a.row_begin = 0;
    a.row_end = (int32_t)n;
    hipLaunchKernelGGL(k3_sym_merge<4>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, a);
    sym_launch_pass2(ctx, a);
    PFZ_HIP(hipGetLastError());
    // the parts' lists -> the result, on every GPU
    const uint64_t *lists = mine.p;
    if (n_parts > 1) {
        PFZ_TRY(comm_allgather_bytes(comm, mine.p, all.p, list_bytes));
        lists = all.p;
    }
    hipLaunchKernelGGL(k3_sym_merge_parts, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, lists, n_parts, (int32_t)n, ntop,
                       inv_scale, out->idx, out->val);
    PFZ_HIP(hipGetLastError());
    s->launches += 1;
    s->rows += n;
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

extern "C" int pfz_index_symmetric_census(const pfz_index *ix, int64_t *magnet_rows, int64_t *recomputed_rows)
{
    PFZ_REQUIRE(ix, "pfz_index_symmetric_census: NULL index");
    int64_t mags = 0, rec = 0;
    const pfz::K3SymState *s = ix->sym;
    if (s && s->n > 0 && s->launches > 0) {
        PFZ_HIP(hipSetDevice(ix->ctx->device));
        std::vector<int32_t> mag((size_t)ix->n_blocks * 32 * pfz::kSymMag);
        int32_t n_ovf = 0;
        PFZ_TRY(pfz::copy_d2h(ix->ctx, mag.data(), s->mag, mag.size() * sizeof(int32_t)));
        PFZ_TRY(pfz::copy_d2h(ix->ctx, &n_ovf, s->ovf, sizeof(int32_t)));
        for (int32_t r : mag) mags += r >= 0;
        rec = n_ovf;
    }
    if (magnet_rows) *magnet_rows = mags;
    if (recomputed_rows) *recomputed_rows = rec;
    return PFZ_OK;
}

extern "C" int pfz_comm_cossim_topn_symmetric(pfz_comm *c, const pfz_index *ix, const pfz_csr *A, int32_t ntop, float lower_bound,
                                              pfz_topn *out)
{
    PFZ_REQUIRE(c && ix && A && out, "pfz_comm_cossim_topn_symmetric: NULL argument");
    PFZ_REQUIRE(ntop >= 1 && lower_bound == lower_bound, "pfz_comm_cossim_topn_symmetric: ntop must be >= 1 and lower_bound a number");
    PFZ_REQUIRE(out->n_rows >= A->n_rows && out->ntop == ntop, "pfz_comm_cossim_topn_symmetric: result buffer is %lldx%d, need %lldx%d",
                (long long)out->n_rows, out->ntop, (long long)A->n_rows, ntop);
    if (!pfz::k3_sym_sharded_ok(ix, A, ntop, pfz::comm_world(c))) {
        pfz::set_error("pfz_comm_cossim_topn_symmetric: not a job for the symmetric form (the matrix must be the one the index was built "
                       "from, 2048-row blocks, at least two of them, top_n <= 32, ranks x top_n <= 256): ask pfz_index_symmetric_ok first");
        return PFZ_ERR_UNSUPPORTED;
    }
    pfz_ctx *ctx = ix->ctx;
    PFZ_HIP(hipSetDevice(ctx->device));
    pfz::ProfScope ps(ctx, "k3_cossim_topn");
    return pfz::k3_sym_sharded(ctx, c, ix, A, ntop, lower_bound, out);
}

extern "C" int pfz_comm_symmetric_ok(pfz_comm *c, const pfz_index *ix, const pfz_csr *A, int32_t ntop, int32_t *yes)
{
    PFZ_REQUIRE(c && ix && A && yes, "pfz_comm_symmetric_ok: NULL argument");
    pfz_ctx *ctx = ix->ctx;
    PFZ_HIP(hipSetDevice(ctx->device));
    // this rank's own answer: the rule of pfz_index_symmetric_ok AND the session buffers of the index (allocated here, so that the
    // job itself has nothing left to fail on that its peers would not see) ...
    bool mine = pfz::k3_sym_sharded_ok(ix, A, ntop, pfz::comm_world(c));
    if (mine) mine = pfz::sym_state_of(ctx, ix) != nullptr;
    // ... and everybody's
    bool all = false;
    PFZ_TRY(pfz::comm_agree(c, mine, &all));
    *yes = all ? 1 : 0;
    return PFZ_OK;
}

extern "C" int pfz_index_symmetric_ok(const pfz_index *ix, const pfz_csr *A, int32_t ntop, int32_t n_parts, int32_t *yes)
{
    PFZ_REQUIRE(ix && A && yes, "pfz_index_symmetric_ok: NULL argument");
    *yes = pfz::k3_sym_sharded_ok(ix, A, ntop, n_parts) ? 1 : 0;
    return PFZ_OK;
}
