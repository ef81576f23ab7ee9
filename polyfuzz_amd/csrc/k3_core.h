// K3's device-side building blocks, shared by the two K3 kernels (k3_cossim_topn.hip: one from-row per wave, to-blocks
// inner; k3_lockstep.hip: all waves on the same to-block, from-rows inner): wave helpers, the candidate buffer of 64-bit
// keys (sum << 32 | ~col) with its compaction, the threshold warm start, the accumulator sweep and the piece-packed
// scatter.  See k3_cossim_topn.hip's header for the data layout and the arithmetic.
// Replaces sparse_dot_topn.awesome_cossim_topn as called at reference polyfuzz/models/_utils.py:82 (+ :84-91, 128-146).
#pragma once
#include "pfz_internal.h"

#include <math.h>
#include <utility>

#ifndef PFZ_K3_EXP
#define PFZ_K3_EXP 0   // timing experiments (tools/build_variant.sh -DPFZ_K3_EXP=n, results wrong on purpose); 0 = the product
#endif

namespace pfz {

constexpr int kSelectMinTop = 16;  // above this top_n, intermediate compactions select instead of sorting
constexpr int kWarmMaxTop = 8;    // threshold warm start (one wave-max round per rank) up to this top_n
constexpr int kPiece = 16;        // postings per piece (one 128-byte line, one 16-lane DPP row)
constexpr long long kPieceBoundMax = 1 << 20;   // pfz_index_build sizes the postings by a bound up to this many pieces (128 MB), by the exact count beyond

// ---------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------
#ifndef PFZ_K3_SHFL_MAX
#define PFZ_K3_SHFL_MAX 0      // 1: the wave maximum by twelve ds_bpermute (the form of rounds 1-4, kept for A/B builds)
#endif
__device__ inline uint64_t wave_max_u64(uint64_t v)
{
#if PFZ_K3_SHFL_MAX
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, d, 64);
        uint32_t hi = __shfl_xor((uint32_t)(v >> 32), d, 64);
        uint64_t o = ((uint64_t)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
#else
    // DPP instead of LDS round trips: the inclusive prefix maximum of dpp_max_scan() below on both halves of the key (row_shr
    // 1, 2, 4, 8 inside each 16-lane row, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3; a lane without a
    // source reads 0, the identity of an unsigned maximum) -- lane 63 ends up with the maximum of the wave, read back as a
    // scalar.  A compaction is `keep` of these in a row, each waiting for the one before: twelve dependent ds_bpermute
    // (~1.5k clocks) against six DPP steps of five instructions.
#define PFZ_MAX_STEP(ctrl, rmask)                                                                           \
    {                                                                                                       \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, ctrl, rmask, 0xf, false);         \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), ctrl, rmask, 0xf, false);  \
        const uint64_t o = ((uint64_t)hi << 32) | lo;                                                       \
        v = o > v ? o : v;                                                                                  \
    }
    PFZ_MAX_STEP(0x111, 0xf)
    PFZ_MAX_STEP(0x112, 0xf)
    PFZ_MAX_STEP(0x114, 0xf)
    PFZ_MAX_STEP(0x118, 0xf)
    PFZ_MAX_STEP(0x142, 0xa)
    PFZ_MAX_STEP(0x143, 0xc)
#undef PFZ_MAX_STEP
    const uint32_t rl = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    const uint32_t rh = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return ((uint64_t)rh << 32) | rl;
#endif
}

// The workgroup is ONE wave: its LDS operations execute in program order, so
// cross-lane hand-offs through LDS need no hardware barrier -- only the compiler
// must not reorder across the hand-off.  (__syncthreads() would also drain vmcnt
// to 0 and stall on every posting load in flight.)
__device__ inline void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // compiler-level ordering, no instruction
    __builtin_amdgcn_wave_barrier();
}

__device__ inline float readlane_f(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

__device__ inline int max3i(int a, int b, int c)
{
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// inclusive prefix maximum over the 64 lanes, unsigned (lane 63 = the wave maximum): row_shr 1, 2, 4, 8 inside each 16-lane
// row, then row_bcast:15 into rows 1, 3 and row_bcast:31 into rows 2, 3; a lane without a source reads 0
__device__ inline uint32_t dpp_umax_scan(uint32_t m)
{
#define PFZ_UMAX_STEP(ctrl, rmask)                                                                    \
    {                                                                                                 \
        const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, ctrl, rmask, 0xf, false); \
        m = o > m ? o : m;                                                                            \
    }
    PFZ_UMAX_STEP(0x111, 0xf)
    PFZ_UMAX_STEP(0x112, 0xf)
    PFZ_UMAX_STEP(0x114, 0xf)
    PFZ_UMAX_STEP(0x118, 0xf)
    PFZ_UMAX_STEP(0x142, 0xa)
    PFZ_UMAX_STEP(0x143, 0xc)
#undef PFZ_UMAX_STEP
    return m;
}

struct TopState {
    int cnt;      // wave-uniform number of keys in cand[]
    int thr;      // accept sum > thr
    int pushed;   // set by every push (k3_lockstep: the row's state has to be written back); never read by the main kernel
};

// Keep the ntop best of cand[0..cnt) sorted at cand[0..keep).
template <int kCap>
__device__ inline void compact(uint64_t *cand, TopState &st, int ntop, int lane, bool sorted = true)
{
    wave_sync();
    constexpr int kPer = (kCap + 63) / 64;   // keys per lane
    uint64_t e[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        int p = lane + 64 * i;
        e[i] = p < st.cnt ? cand[p] : 0ull;
    }
    wave_sync();
    if (!sorted && ntop > kSelectMinTop) {
        // Large top_n, order not needed yet: one wave-max round per kept key (ntop x ~45 instructions) is
        // replaced by a selection -- the ntop-th largest key T is built bit by bit from the top (64 steps
        // of "are there still >= ntop keys >= T | bit ?"), then the keys >= T are written out, unsorted.
        // The keys are distinct (they contain the column), so exactly ntop survive.
        if (st.cnt <= ntop) return;          // nothing to drop, the threshold cannot move
        uint64_t T = 0ull;
        for (int bit = 62; bit >= 0; --bit) {   // sums are < 2^31: bit 63 is never set
            const uint64_t c = T | (1ull << bit);
            int n = 0;
#pragma unroll
            for (int i = 0; i < kPer; ++i) n += __popcll(__ballot(e[i] >= c));
            T = n >= ntop ? c : T;
        }
        int base = 0;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const bool keep_it = e[i] >= T;
            const uint64_t mk = __ballot(keep_it);
            if (keep_it) cand[base + __popcll(mk & ((1ull << lane) - 1ull))] = e[i];
            base += __popcll(mk);
        }
        st.cnt = base;                         // == ntop
        const int t = (int)(uint32_t)(T >> 32) - 1;
        st.thr = t > st.thr ? t : st.thr;
        wave_sync();
        return;
    }
    const int keep = st.cnt < ntop ? st.cnt : ntop;
    // One round per kept key: the wave maximum of the SUMS (upper halves: six v_max_u32 with the DPP operand folded in), then,
    // among the keys with that sum, the maximum of the lower halves (~column: the smallest column wins a tie) -- two 32-bit
    // prefix maxima instead of one 64-bit one (six steps of two DPP moves, a 64-bit compare and two selects: round 4).
    uint32_t hi[kPer], lo[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        hi[i] = (uint32_t)(e[i] >> 32);
        lo[i] = (uint32_t)e[i];
    }
    uint32_t bh = 0, bl = 0;
    for (int r = 0; r < keep; ++r) {
        uint32_t mh = hi[0];
#pragma unroll
        for (int i = 1; i < kPer; ++i) mh = hi[i] > mh ? hi[i] : mh;
        bh = (uint32_t)__builtin_amdgcn_readlane((int)dpp_umax_scan(mh), 63);
        uint32_t ml = 0;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const uint32_t c = hi[i] == bh ? lo[i] : 0u;
            ml = c > ml ? c : ml;
        }
        bl = (uint32_t)__builtin_amdgcn_readlane((int)dpp_umax_scan(ml), 63);
#pragma unroll
        for (int i = 0; i < kPer; ++i)
            if (hi[i] == bh && lo[i] == bl) hi[i] = lo[i] = 0u;      // (keys are distinct: exactly one entry of the wave)
        if (lane == 0) cand[r] = ((uint64_t)bh << 32) | bl;
    }
    st.cnt = keep;
    if (keep == ntop) {
        // from now on only sums >= the ntop-th best can matter
        const int t = (int)bh - 1;
        st.thr = t > st.thr ? t : st.thr;
    }
    wave_sync();
}

// push the entries of one int4 (columns j0..j0+3) that beat the threshold
// (kBounded: ... and whose key is below `ub` -- the deep top-n of pfz_cossim_topn_rows asks for "the next 1024 after this key")
template <int kCap, bool kBounded = false>
__device__ inline void push4(uint64_t *cand, TopState &st, const int4 &v, int j0, int self_col, int ntop, int lane,
                             uint64_t ub = ~0ull)
{
    const int vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j = j0 + c;
        bool pred = vv[c] > st.thr && j != self_col;
        if (kBounded) pred = pred && ((((uint64_t)(uint32_t)vv[c] << 32) | (uint32_t)(~j)) < ub);
        const uint64_t mk = __ballot(pred);
        if (mk) {
            const int pos = st.cnt + __popcll(mk & ((1ull << lane) - 1ull));
            if (pred) cand[pos] = ((uint64_t)(uint32_t)vv[c] << 32) | (uint32_t)(~j);
            st.cnt += __popcll(mk);
            st.pushed = 1;
            if (st.cnt > kCap - 64) compact<kCap>(cand, st, ntop, lane, false);
        }
    }
}

__device__ inline int dpp_max_scan(int m)
{
    // inclusive prefix maximum over the 64 lanes (values >= 0): row_shr 1,2,4,8 inside each 16-lane row,
    // then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x111, 0xf, 0xf, false));
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x112, 0xf, 0xf, false));
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x114, 0xf, 0xf, false));
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x118, 0xf, 0xf, false));
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x142, 0xa, 0xf, false));
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x143, 0xc, 0xf, false));
    return m;
}

// Warm start of the threshold from the first block a from-row touches.  With the threshold still at the
// lower bound every non-zero sum of that block would be pushed (and compacted away again).  The k-th
// largest of the 64 lanes' own maxima is a lower bound of the k-th largest sum of the block (each lane
// maximum is a different to-row), so it can serve as the threshold before the block is filtered.
// Equal maxima are counted once, which only lowers the bound; `k` is ntop, plus one when the self-match
// column is excluded (it may be one of the maxima).
template <int N4>
__device__ inline int warm_threshold(const int4 *acc4, int i_begin, int k, int lane)
{
    int lm = 0;
#pragma unroll
    for (int t = 0; t < N4 / 128; ++t) {
        const int4 v0 = acc4[i_begin + t * 128 + lane], v1 = acc4[i_begin + t * 128 + lane + 64];
        lm = max3i(lm, max3i(v0.x, v0.y, v0.z), max3i(v0.w, v1.x, v1.y));
        lm = max3i(lm, v1.z, v1.w);
    }
    int best = 0;
    for (int r = 0; r < k; ++r) {
        // (the wave maximum of non-negative values: lane 63 of the DPP prefix maximum -- six DPP steps instead of six ds_bpermute)
        best = __builtin_amdgcn_readlane(dpp_max_scan(lm), 63);
        if (best == 0) break;          // fewer than k positive sums
        if (lm == best) lm = 0;
    }
    return best - 1;                   // the filter accepts sum > threshold
}

// Read, clear and filter one block of accumulators: int4 slots [0, N4) of the block whose first column is
// col0.  (The accumulators start at LDS address 0.)
template <int N4, int kCap, bool kBounded = false>
__device__ inline void sweep_block(int4 *acc4, uint64_t *cand, TopState &st, int col0, int self_col, int ntop, int lane,
                                   int zero, uint64_t ub = ~0ull)
{
    static_assert(N4 % 128 == 0, "a wave sweeps whole 128-slot steps");
#pragma unroll 2
    for (int t = 0; t < N4 / 128; ++t) {
        const int i0 = t * 128 + lane, i1 = i0 + 64;
        const int4 v0 = acc4[i0], v1 = acc4[i1];
        // (clearing with eight ds_write_addtid_b32 per step instead -- 2 LDS cycles per 256 B against 13 per
        // KiB -- measured 6 % SLOWER, and reading-and-clearing in one operation, ds_wrxchg2_rtn_b64 with zero, 2 %
        // slower: the sweep is bound by its instruction stream and latencies, not by LDS cycles)
        acc4[i0] = make_int4(zero, zero, zero, zero);
        acc4[i1] = make_int4(zero, zero, zero, zero);
        const int mx = max3i(max3i(v0.x, v0.y, v0.z), max3i(v0.w, v1.x, v1.y), max3i(v1.z, v1.w, v1.w));
        if (__ballot(mx > st.thr)) {
            push4<kCap, kBounded>(cand, st, v0, col0 + i0 * 4, self_col, ntop, lane, ub);
            push4<kCap, kBounded>(cand, st, v1, col0 + i1 * 4, self_col, ntop, lane, ub);
        }
    }
}

// ---------------------------------------------------------------------------
// scatter
// ---------------------------------------------------------------------------
// NS steps of one round.  Lane L = 16q + s holds in (addr_t, as_t) the descriptor of the piece quarter q
// processes in step s: byte offset of the piece in the index and the from-row's scaled value for the
// piece's n-gram.  A step broadcasts lane s of every 16-lane row to the row (DPP row_newbcast, folded
// into the add / the multiply), so quarter q's 16 lanes read the 16 postings of their piece -- one aligned
// 128-byte line -- and apply them: acc[x] += trunc(as * b).  All loads of the round are issued before the
// first is consumed.  The accumulators sit at LDS address 0 (checked in the kernel), so a posting's byte
// offset is its LDS address and the add below folds to nothing.
template <int S> __device__ inline int row_bcast_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + S, 0xf, 0xf, false);   // DPP row_newbcast:S
}

template <int S>
__device__ inline int2 load_piece_entry(const char *__restrict__ post_bytes, int addr_t, int sub8)
{
    const uint32_t a = (uint32_t)(row_bcast_i<S>(addr_t) + sub8);
#if PFZ_K3_EXP == 2      // no posting loads
    return make_int2((int)((a * 2654435761u) >> 19) & 8188, 0x3c000000);
#else
    return *(const int2 *)(post_bytes + a);
#endif
}

template <int S> __device__ inline void apply_entry(int *acc, const int2 &pe, float as_t)
{
    const int v = (int)(__int_as_float(row_bcast_i<S>(__float_as_int(as_t))) * __int_as_float(pe.y));
#if PFZ_K3_EXP == 1      // no LDS atomics
    if (v == 0x7fffffff) atomicAdd((int *)((char *)acc + pe.x), v);
#elif PFZ_K3_EXP == 3    // conflict-free LDS addresses
    atomicAdd((int *)((char *)acc + ((pe.x & 0x1f00) | (threadIdx.x * 4))), v);
#else
    atomicAdd((int *)((char *)acc + pe.x), v);                               // ds_add_u32, no return
#endif
}

template <int... S>
__device__ inline void run_steps_seq(int *acc, const char *__restrict__ post_bytes, int addr_t, float as_t, int sub8,
                                     std::integer_sequence<int, S...>)
{
    const int2 pe[sizeof...(S)] = {load_piece_entry<S>(post_bytes, addr_t, sub8)...};
    (apply_entry<S>(acc, pe[S], as_t), ...);
}

template <int NS>
__device__ inline void run_steps(int *acc, const char *__restrict__ post_bytes, int addr_t, float as_t, int sub8)
{
    run_steps_seq(acc, post_bytes, addr_t, as_t, sub8, std::make_integer_sequence<int, NS>{});
}

// inclusive prefix sum over the 64 lanes: six v_add_u32_dpp (row_shr 1, 2, 4, 8 inside each 16-lane row, then row_bcast 15 / 31
// across the rows).  Written out: from `x += update_dpp(0, x, ..)` the compiler made a move of the identity, a DPP move and an add
// per step, and kept the moved partials alive to re-derive the exclusive sum from them -- 26 vector instructions per (row, block)
// of a kernel that issues them in 60 % of its slots.  (s_nop 1: a DPP operand written by the instruction before needs two wait
// states, and the hazard is not looked for between inline statements.)
__device__ inline int dpp_add_scan(int v)
{
    asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
    asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
    asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
    asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
    asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    return v;
}

// Scatter the (k,b) lists of up to 64 n-grams of one from-row into the accumulators.
//   np, st : per lane, number of pieces and first piece of the lane's list in this block (np == 0: nothing)
//   as     : per lane, the row's value for the lane's n-gram times the fixed-point scale
//   mark   : 64 ints of LDS scratch (the unused tail of the candidate buffer)
// The pieces of all lists are numbered P = 0..T-1 (prefix sum of np) and taken 64 per round; in a round
// quarter q of the wave processes pieces 4s + q, s = 0.., so the four quarters stay busy to the last step
// whatever T is.  Owner search, once per round: every list that overlaps the round's window writes its lane
// number at the window position of its first piece, a prefix maximum spreads it over the list's pieces.
__device__ inline void scatter_pieces(int *acc, const char *__restrict__ post_bytes, int *mark, int np, int st, float as,
                                      int lane, int src4, int sub8, int dummy_addr)
{
    // inclusive scan over the 64 lanes with DPP adds (no LDS traffic, unlike __shfl_up/ds_bpermute)
    const int pin = dpp_add_scan(np);
    const int total = __builtin_amdgcn_readlane(pin, 63);
    const int excl = pin - np;
    const int base = st - excl;            // piece P of this list is index piece base + P
    const int pt = 4 * (lane & 15) + (lane >> 4);   // window position of the piece this lane processes
    for (int r0 = 0; r0 < total; r0 += 64) {
#if PFZ_K3_EXP == 4      // no owner search: every lane takes a piece of its own list
        const int addr_t = np > 0 ? (int)((uint32_t)(st + ((r0 + pt) % np)) << 7) : dummy_addr;
        const float as_t = as;
#else
        const int rel = excl - r0;
        mark[lane] = 0;
        if (np > 0 && rel < 64 && rel + np > 0) mark[rel > 0 ? rel : 0] = lane + 1;
        wave_sync();
        const int owner1 = dpp_max_scan(mark[lane]);                         // owner lane + 1 of window position `lane`
        const int o4 = __builtin_amdgcn_ds_bpermute(src4, owner1) * 4 - 4;   // ... of window position pt, as a bpermute index
        const int b_o = __builtin_amdgcn_ds_bpermute(o4, base);
        const float as_t = __int_as_float(__builtin_amdgcn_ds_bpermute(o4, __float_as_int(as)));
        const int P = r0 + pt;
        const int addr_t = P < total ? (int)((uint32_t)(b_o + P) << 7) : dummy_addr;
        wave_sync();                                                         // mark[] is rewritten by the next round
#endif
        const int left = total - r0;
        // groups of two steps (8 pieces); an explicit binary tree of wave-uniform branches (the compiler lowers a
        // switch to a chain of compares with saved/restored condition masks -- ~40 scalar instructions per round)
        const int ng = left >= 64 ? 8 : (left + 7) >> 3;
#if PFZ_K3_EXP == 6      // groups of four steps only
        if (ng <= 4) {
            if (ng <= 2) run_steps<4>(acc, post_bytes, addr_t, as_t, sub8);
            else run_steps<8>(acc, post_bytes, addr_t, as_t, sub8);
        } else {
            if (ng <= 6) run_steps<12>(acc, post_bytes, addr_t, as_t, sub8);
            else run_steps<16>(acc, post_bytes, addr_t, as_t, sub8);
        }
#else
        if (ng <= 4) {
            if (ng <= 2) {
                if (ng == 1) run_steps<2>(acc, post_bytes, addr_t, as_t, sub8);
                else run_steps<4>(acc, post_bytes, addr_t, as_t, sub8);
            } else {
                if (ng == 3) run_steps<6>(acc, post_bytes, addr_t, as_t, sub8);
                else run_steps<8>(acc, post_bytes, addr_t, as_t, sub8);
            }
        } else {
            if (ng <= 6) {
                if (ng == 5) run_steps<10>(acc, post_bytes, addr_t, as_t, sub8);
                else run_steps<12>(acc, post_bytes, addr_t, as_t, sub8);
            } else {
                if (ng == 7) run_steps<14>(acc, post_bytes, addr_t, as_t, sub8);
                else run_steps<16>(acc, post_bytes, addr_t, as_t, sub8);
            }
        }
#endif
    }
}

// ---- host side: K3's fixed-point scale ------------------------------------------------------------------------------
// |sum| <= ||a|| * ||b|| (Cauchy-Schwarz) must stay below 2^31: S = 2^k, thr0 = floor(lower_bound * S) (accept sum > thr0)
inline void k3_fixed_point(const pfz_csr *A, const pfz_index *ix, float lower_bound, float *scale, float *inv_scale, int32_t *thr0)
{
    if (lower_bound < 0.f) lower_bound = 0.f;
    const double bound = (double)A->max_norm * (double)ix->max_norm * 1.0001 + 1e-30;
    int k = 30;
    while (k > -60 && ldexp(bound, k) >= 2147483000.0) --k;
    while (k < 60 && ldexp(bound, k + 1) < 1073741824.0) ++k;   // tiny norms: use the full range
    *scale = (float)ldexp(1.0, k);
    *inv_scale = (float)ldexp(1.0, -k);
    const double thr_d = floor((double)lower_bound * (double)*scale);
    *thr0 = thr_d >= 2147483000.0 ? 2147483000 : (int32_t)thr_d;
}

// ---- host side, k3_lockstep.hip ----------------------------------------------------------------------------------
// The to-block-major ("lock-step") form of K3 for to-sides whose index outgrows the L2s: true when it should run.
bool k3_lockstep_wanted(const pfz_ctx *ctx, const pfz_index *ix, int64_t n_rows, int32_t ntop);
int k3_lockstep_launch(pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int64_t row_begin, int64_t n_rows, int32_t ntop,
                       int32_t thr0, float scale, float inv_scale, int32_t exclude_diag, int64_t diag_offset, pfz_topn *out);


// ---- host side, k3_symmetric.hip ---------------------------------------------------------------------------------
// The symmetric form of K3 for a self-match (every unordered pair scored once).  k3_sym_wanted: 0 = not a job for it,
// 1 = start a session with this row range, 2 = the range continues the running session.
int k3_sym_wanted(const pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int64_t row_begin, int64_t row_end, int32_t ntop,
                  int32_t thr0, float scale, int32_t exclude_diag, int64_t diag_offset, const pfz_topn *out);
int k3_sym_launch(pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int64_t row_begin, int64_t row_end, int32_t ntop,
                  int32_t thr0, float scale, float inv_scale, pfz_topn *out, bool start, bool *declined);
int k3_sym_launch_streamed(pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int32_t ntop, int32_t thr0, float scale, float inv_scale,
                           pfz_topn *out, int32_t n_ranges, const int64_t *ends, int32_t first_event, int32_t *host_idx, float *host_val,
                           bool *declined);
void k3_sym_free(pfz_index *ix);

}  // namespace pfz
