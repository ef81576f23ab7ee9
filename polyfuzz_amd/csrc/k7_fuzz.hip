// K7: the rapidfuzz.fuzz scorers that build a different string pair for every (from, to) -- partial_ratio,
// token_set_ratio, token_ratio, partial_token_*_ratio and WRatio, the default scorer of the reference's RapidFuzz
// matcher (polyfuzz/models/_rapidfuzz.py:45-58, 106-108) -- all pairs + process.extractOne's first best choice.
// (ratio / QRatio / token_sort_ratio are one fixed string per list element: K4, k4_indel.hip.)
//
// The arithmetic of one pair -- exact score, and a cheap upper bound of it -- is k7_core.h; the preparation of the lists (the
// first two items below) is k7_plan.hip / k7_plan.h; this file is the match kernel and its launch.  All of it on the device:
//
//  * the three FORMS of a list (the strings; their whitespace tokens sorted and joined; their distinct tokens sorted and
//    joined, with the tokens' positions / lengths / hashes) -- k7_tokenize, one thread per string, cached on the list's
//    pfz_strings handle: a property of the list alone;
//  * the to-side PLAN, cached on the to-list's handle: alphabet (ranks of the distinct code points of the to-list plus
//    the joining space) and character classes by frequency, a hash table of the to-list's distinct tokens (token id =
//    the table's representative: equal ids <=> equal tokens), to-strings sorted by length into groups of 64 with
//    everything a lane needs stored [position][lane] -- symbols of the three forms, token tags, token ids / lengths,
//    lengths, class histogram, token signature (k7_pack);
//  * per call: the from-list's token ids by look-up in the plan's table, then the match kernel.
//
// The match kernel (k7_fuzz_kernel<W>): persistent one-wave workgroups (16 per CU for W = 1) take units -- a from-string,
// or a share of one -- from an atomic counter; the wave holds the bit-parallel match tables of the from-string's three
// forms in LDS (W 64-bit words per form).  process.extractOne keeps only the maximum, so almost every pair can be
// dismissed without scoring it:
//    sweep 1  every lane walks its to-strings computing only the UPPER BOUND of the pair's score (lengths, common
//             character classes by v_sad_u8, token signatures and the first four token ids), leaves it as a byte in the
//             workgroup's stretch of the bound cache, and remembers its best-bounded one: those 64 SEEDS are scored
//             exactly -- the best of them is `cur`, a score some valid choice really has;
//    sweep 2  a tight loop over the bytes: a pair whose bound (+ slack) is below `cur` cannot be the answer, nor tie with
//             it; the survivors are compacted (ballot) into a queue and scored 64 at a time, one pair per lane, each
//             raising `cur`;
//    window sweeps (partial_ratio and what builds on it) are not part of a pair's scoring: the pair asks for them, runs
//             of windows are handed to lanes of their own (sweep_rounds);
//    heavy rows -- known from a count of the bytes that reach the seeds' best -- are dealt to continuation units, which
//             other workgroups take when the rows run out (FuzzArgs::cont_list).
// Pruned pairs are strictly worse than the final best, so the first-best rule (score desc, original index asc) over the
// scored pairs is the exact answer.  On the 20 000 x 20 000 IMDB title lists 6 % of the pairs are scored (WRatio).
// How it got here, with the measurements: DESIGN.md, "K7".
//
// From-strings beyond 256 characters or 32 distinct tokens (and to-strings beyond 32 distinct tokens, for every
// from-string) take k7_general_kernel: the same scorers with any number of words and tokens, all state in global
// scratch -- slow, but the reference accepts such inputs.   PARITY UNPINNED (rapidfuzz is not installable): the oracle
// is oracle/fuzz_scorers.{py,c}, anchored on rapidfuzz's published values.
#include "pfz_internal.h"

#include <algorithm>
#include <climits>

#undef PFZ_HD
#define PFZ_HD __device__ inline
#define PFZ_LDS_U16 __attribute__((address_space(3))) uint16_t
#define PFZ_LDS_U8 __attribute__((address_space(3))) uint8_t
#ifdef PFZ_K7_PROFILE
#define FZ_TICK(T, k)                                                 \
    do {                                                              \
        const long long now_ = clock64();                             \
        (T).tk[k] += (unsigned int)(now_ - (T).t0);                   \
        (T).t0 = now_;                                                \
    } while (0)
#endif
// (the builtin takes the condition as it is: __ballot() compares an integer copy of it with 0 -- a select and a compare more)
#define FZ_BALLOT(x) __builtin_amdgcn_ballot_w64(x)
#define FZ_ANY(x) (FZ_BALLOT(x) != 0ull)
#include "k7_core.h"
#include "k7_args.h"
#include "k7_plan.h"

namespace pfz {

// One from-string per WAVE (a one-wave workgroup): what a from-string costs on top of its pairs -- the bounding sweeps, a
// batch of seeds, a last partial batch -- is paid per wave that works on it, and four waves sharing one from-string paid
// it four times over (the 100 000-name self-match: 0.79 -> 0.49 s).  More waves per workgroup = fewer from-strings in flight.
constexpr int kK7Waves = 1, kK7Threads = 64 * kK7Waves;
#ifndef PFZ_K7_OCC
#define PFZ_K7_OCC 4          // one-word rows: waves per SIMD the compiler budgets registers for (tuning builds: 3 = no spills)
#endif
constexpr float kBoundSlack = 0.05f;
// log2 of the windows per run of a window sweep: 16, more for long forms -- at most 8 runs (forms are <= 256 symbols here)
#ifndef PFZ_K7_SHARE_LOG2
#define PFZ_K7_SHARE_LOG2 6
#endif
__device__ inline int sweep_share_log2(int n_windows)
{
    constexpr int k = PFZ_K7_SHARE_LOG2;
    return n_windows <= (8 << k) ? k : (n_windows <= (16 << k) ? k + 1 : (n_windows <= (32 << k) ? k + 2 : k + 3));
}
// heavy-row hand-over (see FuzzArgs::cont_list): a row that has scored kHandBatches batches and still has kHandMinGroups
// groups to go leaves them to continuation units -- as many (up to kContParts) as leave each about kHandBatches batches
constexpr int kHandBatches = 64, kHandMinGroups = 16, kContParts = 64;
constexpr int kHandShortLen = 8, kHandShortBatches = 24;      // ... a from-string of up to 8 characters from 24 batches on (round 4, with the presence bound: 6 -> 8)
static bool mode_sweeps_windows(int mode) { return mode != kTokenSetRatio && mode != kTokenRatio; }

// ---- the match kernel -----------------------------------------------------------------------------------------------------


struct RowBest {
    double score;
    int idx;
    __device__ void take(double s, int i)
    {
        if (s > score || (s == score && i < idx)) {
            score = s;
            idx = i;
        }
    }
};

__device__ inline void wave_best(RowBest &b)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double os = __shfl_xor(b.score, d, 64);
        const int oi = __shfl_xor(b.idx, d, 64);
        b.take(os, oi);
    }
}

// how many of the lanes below this one are set in a wave mask (v_mbcnt: two instructions)
__device__ inline int lanes_below(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// v where any of m's four components equals id (a wave-uniform value, in a scalar register), else 0
__device__ inline int select_if_any_eq(const int4 &m, int id, int v)
{
    uint64_t e0, e1, e2, e3;
    int r;
    asm("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(e0) : "s"(id), "v"(m.x));
    asm("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(e1) : "s"(id), "v"(m.y));
    asm("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(e2) : "s"(id), "v"(m.z));
    asm("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(e3) : "s"(id), "v"(m.w));
    const uint64_t any = (e0 | e1) | (e2 | e3);
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(any));
    return r;
}

__device__ inline bool mode_uses_tokens(int mode) { return mode != kPartialRatio && mode != kPartialTokenSortRatio; }

// 64-bit word-steps a scored pair costs at most (work accounting for the roofline; an estimate from the lengths: one
// pass over the to-form per LCS, |from| x |to| for a window sweep)
__device__ inline int work_estimate(const Fz3<int> &la, const int4 &m, int mode, int W)
{
    const int pass = m.x + m.y + m.z;
    if (mode == kTokenSetRatio || mode == kTokenRatio) return pass * W;
    return (pass + fz_min(la[0], m.x) * fz_max(la[0], m.x) / 4) * W;
}

template <int W, int MODE = -1>
__global__ __launch_bounds__(kK7Threads, W == 1 ? PFZ_K7_OCC : (W == 2 ? 3 : 2)) void k7_fuzz_kernel(FuzzArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // dynamic part: the from-string's match table [symbol][form][word]; the scratch columns of the window sweeps
    // [position][lane], bytes when the alphabet has at most 255 symbols; (profiling) the phase timers
    uint64_t *pm = (uint64_t *)smem_raw;
    const bool narrow = A.n_sym1 <= 256;
    unsigned char *s_stage = smem_raw + (size_t)A.n_sym1 * 3 * W * sizeof(uint64_t);
    unsigned long long *s_ticks = (unsigned long long *)(s_stage + (size_t)kK7Waves * kFuzzStage * 64 * (narrow ? 1 : 2));
    __shared__ int s_la[3], s_ta, s_nspace, s_usum;
    __shared__ int s_tid[kFuzzMaxTokens], s_tlen[kFuzzMaxTokens];
    __shared__ uint64_t s_tmask[kFuzzMaxTokens * W], s_smask[kFuzzMaxTokens * W];
    __shared__ int s_cnt[4 * kFuzzHistWords];
    __shared__ uint32_t s_hist[kFuzzHistWords], s_sig[2];
    __shared__ uint32_t s_pres[2];
    __shared__ double red_s[kK7Waves];
    __shared__ int red_i[kK7Waves];
    __shared__ unsigned long long s_best;          // bits of the best score any lane of the workgroup has found so far (>= 0)
    __shared__ int s_queue[kK7Waves][128];
    __shared__ int s_sweeps[kK7Waves][3][64];     // runs of windows waiting for a lane (see sweep_rounds): item, lengths, symbols
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mode = MODE >= 0 ? MODE : A.mode, parts = A.parts;      // (MODE >= 0: the scorer a compile-time constant)
    const bool use_tokens = mode_uses_tokens(mode);
    const bool use_pres = mode != kTokenSetRatio && mode != kTokenRatio;      // the symbol-presence term of the bound (see bound_of)

    for (int p = tid; p < A.n_sym1 * 3 * W; p += kK7Threads) pm[p] = 0ull;
    __syncthreads();

    __shared__ int s_unit, s_cont[4];
    // profiling only: where the waves' time goes -- 0 set-up of a from-string, 1 / 2 the two bounding sweeps, 3 scoring,
    // 4 the end of a unit (merge, table clean-up, next unit), 5 waiting for a continuation record
    if (A.phase_ticks && tid < 24) s_ticks[tid] = 0ull;
    long long t_last = A.phase_ticks ? clock64() : 0;
    auto tick = [&](int k) {
        if (A.phase_ticks) {
            const long long now = clock64();
            if (tid == 0) s_ticks[k] += (unsigned long long)(now - t_last);
            t_last = now;
        }
    };
    // the workgroup's stretch of the bound cache: its own, until a row it hands over takes that stretch along (the row's
    // continuation units walk the bytes sweep 1 left there) and the workgroup moves on to a spare one
    int my_region = blockIdx.x;
    for (;;) {
        // the from-strings differ by orders of magnitude in how many pairs survive their bound: units are handed out one at
        // a time (an atomic counter) instead of by a fixed stride, so no workgroup is left with a run of heavy ones
        if (tid == 0) s_unit = atomicAdd(A.next_unit, 1);
        __syncthreads();
        const int u = s_unit;
        const int n_primary = A.n_rows * parts;
        const bool is_cont = u >= n_primary;
        tick(4);
        const long long t_begin = A.row_stats ? wall_clock64() : 0;
        unsigned int n_bounded = 0, n_scored = 0, n_steps = 0;        // (work counters of this unit, per lane)
        int r, part, g_first, g_step, cont_rec = -1;
        unsigned long long cur0 = 0ull;
        if (!is_cont) {
            r = u / parts;
            part = u - r * parts;
            g_first = wave + kK7Waves * part;
            g_step = kK7Waves * parts;
        }
        else {
            // a share of a heavy row: the row's hand-over left one record per continuation unit {row, first group, step, units
            // | part << 8 (-1: void)}; unit numbers are dealt in the order the records were claimed.  The record may not be
            // there yet (its row is still being worked on), or never come (all rows finished: the list is final)
            if (!A.cont_list) break;
            const int c = u - n_primary;
            if (tid == 0) {
                // (relaxed polls with long sleeps: an acquire per poll would invalidate the caches the working waves live on)
                int ok;
                for (;;) {
                    if (c < min(__hip_atomic_load(A.n_cont, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), A.cont_cap)) {
                        ok = 1;
                        break;
                    }
                    if (__hip_atomic_load(A.rows_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_primary) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        ok = c < min(__hip_atomic_load(A.n_cont, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), A.cont_cap);
                        break;
                    }
                    for (int z = 0; z < 16; ++z) __builtin_amdgcn_s_sleep(127);
                }
                if (ok) {
                    while (__hip_atomic_load(&A.cont_list[c].w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(64);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                s_cont[0] = ok ? __hip_atomic_load(&A.cont_list[c].w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                if (ok) {
                    s_cont[1] = __hip_atomic_load(&A.cont_list[c].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_cont[2] = __hip_atomic_load(&A.cont_list[c].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_cont[3] = __hip_atomic_load(&A.cont_list[c].z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
            tick(5);
            if (!s_cont[0]) break;
            if (s_cont[0] < 0) continue;           // (a void record: its row could not hand over)
            part = s_cont[0] >> 8;
            r = s_cont[1];
            g_first = s_cont[2];
            g_step = s_cont[3];
            cont_rec = c - part;                   // (the row's first record: where its units share their best score)
            cur0 = __hip_atomic_load(&A.cont_cur[cont_rec], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int row = A.rows[r];
        const int64_t a0 = A.a_off[row];
        // ---- the from-string's tables, class histogram, tokens
        if (tid < 4 * kFuzzHistWords) s_cnt[tid] = 0;
        if (tid < 2) s_pres[tid] = 0u;
        if (tid == 0) {
            s_nspace = 0;
            s_best = cur0;
        }
        __syncthreads();
        for (int v = 0; v < 3; ++v) {
            const int m = v == 0 ? (int)(A.a_off[row + 1] - a0) : (v == 1 ? A.a_len1[row] : A.a_len2[row]);
            if (tid == 0) s_la[v] = m;
            for (int p = tid; p < m; p += kK7Threads) {
                const uint32_t c = load_unit(A.a_form[v], A.a_width, a0 + p);
                const int sy = c < A.lut_len ? (int)A.lut[c] : 0;
                if (sy) {
                    atomicOr((unsigned long long *)&pm[(sy * 3 + v) * W + (p >> 6)], 1ull << (p & 63));
                    if (v == 0) {
                        atomicAdd(&s_cnt[A.cls[sy]], 1);
                        if (sy == A.space_rank) atomicAdd(&s_nspace, 1);
                        if (!is_space_cp(c)) atomicOr(&s_pres[(sy & 63) >> 5], 1u << (sy & 31));      // (no whitespace symbol: see k7_pack)
                    }
                }
            }
        }
        if (tid == 0) {
            const int64_t tb = tok_base(a0, row);
            const int ta = A.a_ntok[row];
            s_ta = ta;
            int start = 0;
            uint64_t sig = 0ull;
            for (int i = 0; i < ta; ++i) {
                const int len = A.a_tok_len[tb + i];
                s_tid[i] = A.a_tok_id[tb + i];
                s_tlen[i] = len;
                sig |= fz_sig_bit(s_tid[i]);
                uint64_t tm[W], sm[W];
                fz_range_mask<W>(tm, start, start + len);
                fz_range_mask<W>(sm, start + len, i + 1 < ta ? start + len + 1 : start + len);
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    s_tmask[i * W + w] = tm[w];
                    s_smask[i * W + w] = sm[w];
                }
                start += len + 1;
            }
            s_sig[0] = (uint32_t)sig;
            s_sig[1] = (uint32_t)(sig >> 32);
        }
        __syncthreads();
        if (tid == 0) {
            const int nt_all = A.a_ntok_all[row];
            const int extra = (nt_all - 1) - s_nspace;
            if (extra > 0) s_cnt[A.space_class] += extra;
            int usum = 0, big = 0;
            for (int c = 0; c < 4 * kFuzzHistWords; ++c) {
                usum += s_cnt[c];
                big |= s_cnt[c] > 255;
            }
            for (int d = 0; d < kFuzzHistWords; ++d)
                s_hist[d] = (uint32_t)s_cnt[4 * d] | (uint32_t)s_cnt[4 * d + 1] << 8 | (uint32_t)s_cnt[4 * d + 2] << 16 | (uint32_t)s_cnt[4 * d + 3] << 24;
            s_usum = big ? -1 : usum;
        }
        __syncthreads();
        FuzzFrom<W> F;
        F.pm = pm;
        F.la = {s_la[0], s_la[1], s_la[2]};
        F.ta = s_ta;
        F.tid = s_tid;
        F.tlen = s_tlen;
        F.tmask = s_tmask;
        F.smask = s_smask;
        FuzzSummary sa;
#pragma unroll
        for (int v = 0; v < 3; ++v) sa.len[v] = F.la[v];
        sa.ntok = F.ta;
#pragma unroll
        for (int d = 0; d < kFuzzHistWords; ++d) sa.hist[d] = s_hist[d];
        sa.usum = s_usum;
        sa.sig = (uint64_t)s_sig[0] | (uint64_t)s_sig[1] << 32;
        const uint32_t pres_a0 = s_pres[0], pres_a1 = s_pres[1];
        const int n_pres_a = __popc(pres_a0) + __popc(pres_a1);
        const int skip = A.skip_idx ? A.skip_idx[row] : -1;
        // choice_left_out(w, skip, up_to) for w >= 0 as ONE unsigned range test (sweep 1 runs it for every pair): the choices
        // skip_lo .. skip_lo + skip_span are left out -- "equal": skip alone; "up to": 0 .. skip; none (-1): the empty range at -1
        const int skip_lo = A.skip_up_to && skip >= 0 ? 0 : skip;
        const unsigned skip_span = (unsigned)(skip - skip_lo);

        auto cur_now = [&]() {
            return __longlong_as_double((long long)__hip_atomic_load(&s_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        };
        auto to_of = [&](int slot, const int4 &m) {
            const int4 rec = A.b_meta3[slot];
            FuzzTo T;
            T.sym = {A.b_sym + rec.x, A.b_sym + rec.x + rec.w, A.b_sym + rec.x + 2 * rec.w};
            T.tag = A.b_tag + rec.y;
            T.tok_id = A.b_tok_id + rec.z;
            T.tok_len = A.b_tok_len + rec.z;
            T.lb = {m.x, m.y, m.z};
            T.tb = m.w;
            T.stage = nullptr;
            T.stage_stride = 0;          // (the scratch columns belong to the window sweeps: sweep_rounds)
            T.staged = -1;
            T.n_windows = 0;
            return T;
        };
        // what a sweep reads of one to-string: five 128-bit loads, all issued one group ahead of their use
        struct Meta {
            int4 m, m2, m4;
            uint4 h0, h1;
            uint2 p;
        };
        // (byte offsets in 32 bits: a scalar base + one vector offset per load instead of 64-bit address arithmetic per lane
        // and array -- the to-side holds at most 2^26 strings, 2^30 B of its widest array)
        auto load_meta = [&](int g) {
            const uint32_t off = ((uint32_t)g * 64u + (uint32_t)lane) * 16u;
            Meta x;
            x.m = *(const int4 *)((const char *)A.b_meta + off);
            x.m2 = *(const int4 *)((const char *)A.b_meta2 + off);
            x.m4 = *(const int4 *)((const char *)A.b_meta4 + off);
            const uint32_t hoff = ((uint32_t)g * 128u + (uint32_t)lane) * 16u;
            x.h0 = *(const uint4 *)((const char *)A.b_hist + hoff);
            x.h1 = *(const uint4 *)((const char *)A.b_hist + hoff + 1024u);
            x.p = use_pres ? *(const uint2 *)((const char *)A.b_pres + (off >> 1)) : make_uint2(0u, 0u);
            return x;
        };
        // float32 upper bound of the pair (from-string, to-string x) FROM REGISTERS ONLY; valid = a real candidate of this
        // kernel.  When the token signatures meet, the exact intersection is taken from the first four token ids (nearly
        // every to-string has no more); `coarse` = the signatures meet but the string has more tokens: the bound assumes
        // the best, and the pair is looked at again -- with its record -- when it is popped from the queue.
        auto summary_of = [&](const Meta &x) {
            FuzzSummary sb;
            sb.len[0] = x.m.x;
            sb.len[1] = x.m.y;
            sb.len[2] = x.m.z;
            sb.ntok = x.m.w;
            sb.hist[0] = x.h0.x;
            sb.hist[1] = x.h0.y;
            sb.hist[2] = x.h0.z;
            sb.hist[3] = x.h0.w;
            sb.hist[4] = x.h1.x;
            sb.hist[5] = x.h1.y;
            sb.hist[6] = x.h1.z;
            sb.hist[7] = x.h1.w;
            sb.usum = x.m2.z;
            sb.sig = (uint64_t)(uint32_t)x.m2.x | (uint64_t)(uint32_t)x.m2.y << 32;
            return sb;
        };
        auto bound_of = [&](const Meta &x, bool &valid, bool &coarse) -> float {
            const int4 m = x.m;
            valid = x.m2.w >= 0 && (unsigned)(x.m2.w - skip_lo) > skip_span && m.w <= kFuzzMaxTokens;
            const FuzzSummary sb = summary_of(x);
            const int uu = fz_common_chars(sa, sb);
            const bool maybe = use_tokens && (sa.sig & sb.sig) != 0ull;
            coarse = maybe && m.w > 4;
            // the common tokens: every lane compares its (up to four) to-token ids with the from-string's, which sit in
            // scalar registers -- one pass, no branch (disjoint signatures simply find nothing)
            // (the four compares leave lane masks in scalar registers, or-ed there; one select and one add per from-token carry
            // both sums -- common tokens << 16 | their characters, a form is shorter than 2^16: 8 vector instructions per token
            // where the compiler's own rendering of `hit ? .. : ..` took 18, and the kernel is bound by its vector issue rate)
            int nc = 0, sect_chars = 0;
            if (use_tokens) {
                int acc = 0;
                for (int i = 0; i < F.ta; ++i) {
                    const int ida = __builtin_amdgcn_readfirstlane(s_tid[i]);          // (absent to-tokens are -1, unknown from-tokens <= -2)
                    acc += select_if_any_eq(x.m4, ida, s_tlen[i] | 0x10000);
                }
                nc = acc >> 16;
                sect_chars = acc & 0xffff;
            }
            // symbols of the one string that the other lacks (fz_presence_miss, the from-side's bits in scalar registers)
            // (only where it pays: the scorers that sweep windows, WRatio included -- 20k x 20k titles, WRatio 8.38 -> 8.04 ms
            // with 3.8 % of the pairs scored instead of 6.0 %; token_ratio, which sweeps none, 5.33 -> 5.86 ms for 2.5 %
            // instead of 2.9 %: there the bits stay out, a wave-uniform choice)
            int miss_a = 0, miss_b = 0;
            if (use_pres) {
                const int pres_common = __popc(pres_a0 & x.p.x) + __popc(pres_a1 & x.p.y);
                miss_a = n_pres_a - pres_common;
                miss_b = __popc(x.p.x) + __popc(x.p.y) - pres_common;
            }
            // (WRatio reads the token-set bound in one of its two families only -- fz_upper_bound; groups are sorted by length,
            // so whole waves skip it)
            const bool wants_tset = mode == kTokenSetRatio || mode == kTokenRatio ||
                                    (mode == kWRatio && 2 * max(sa.len[0], m.x) < 3 * min(sa.len[0], m.x));
            const float tset = nc != 0 && wants_tset ? fz_token_set_bound_n(F.la[2], F.ta, m.z, m.w, uu, nc, sect_chars, miss_a, miss_b) : -1.0f;
            return fz_upper_bound(sa, sb, mode, uu, coarse ? -1 : (nc != 0 ? 1 : 0), coarse ? -1.0f : tset, miss_a, miss_b);
        };
        RowBest best = {-1.0, INT_MAX};
        tick(0);
        // ---- window sweeps.  Few of the scored pairs need them, those that do need many times the work of the others,
        // and how many windows a pair sweeps is anybody's guess: swept inside the scoring batches, one lane in ten would
        // work and the wave would wait for the slowest.  So scoring only ASKS for sweeps: the windows of a form are shared
        // out in runs of kSweepShare, every run an item of a ring; a lane takes an item, sweeps one window per round of the
        // wave, and takes the next item when its run is done -- rounds are held only while all 64 lanes have a run (or at
        // the end of a sweep of the to-list), a lane's unfinished run waits in its registers in between.
        int sw_item = -1;                       // this lane's run (-1: none): slot | form << 26 | run << 28
        // (what a run keeps between rounds, packed: lb | lb0 << 16, w | w_end << 16, bl | bs << 16 -- lengths <= 256 W)
        int sw_len = 0, sw_sym = 0, sw_win = 0, sw_best = 1 << 16;      // (sw_sym: where the form's symbols are, from b_sym)
        int (*sweeps)[64] = s_sweeps[wave];
        int sq_head = 0, sq_tail = 0;
        auto sweep_rounds = [&](bool flush) {
            for (;;) {
                const unsigned long long idle = FZ_BALLOT(sw_item < 0);
                const int n_take = min((int)__popcll(idle), sq_tail - sq_head);
                const int n_busy = 64 - (int)__popcll(idle) + n_take;
                // (the lane's column: [position][lane] elements of 1 or 2 bytes)
                PFZ_LDS_U16 *column = (PFZ_LDS_U16 *)(s_stage + (size_t)wave * kFuzzStage * 64 * (narrow ? 1 : 2) + lane * (narrow ? 1 : 2));
                // every waiting run is taken whenever the lanes are not all busy (the ring holds 64): a round is held only
                // when all of them are, or at the end
                if (sw_item < 0) {
                    const int rank = lanes_below(idle);
                    if (rank < n_take) {
                        // (the pair's scoring lane left lengths and the form's place beside the item: the lane goes to memory
                        // once, for the symbols)
                        const int at = (sq_head + rank) & 63;
                        sw_item = sweeps[0][at];
                        sw_len = sweeps[1][at];
                        sw_sym = sweeps[2][at];
                        const int v = (sw_item >> 26) & 3, run = sw_item >> 28, lb = sw_len & 0xffff;
                        const int n_win = fz_n_windows(F.la[v], lb), sh = sweep_share_log2(n_win);
                        sw_win = (run << sh) | min((run + 1) << sh, n_win) << 16;
                        sw_best = 1 << 16;
                        if (lb <= kFuzzStage) fz_stage_form(A.b_sym + sw_sym, lb, column, 64, narrow);
                    }
                }
                sq_head += n_take;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (n_busy == 0 || (!flush && n_busy < 64)) break;
                if (sw_item >= 0) {
                    const int slot = sw_item & 0x3ffffff, v = (sw_item >> 26) & 3;
                    const int lb = sw_len & 0xffff, lb0 = sw_len >> 16;
                    const bool staged = lb <= kFuzzStage;
                    const uint16_t *sym = A.b_sym + sw_sym;      // (a long to-form is read where it is)
                    FuzzSweep SW;
                    fz_sweep_begin(SW, v, F.la[v], lb, sw_win & 0xffff, sw_win >> 16, sym, column, staged ? 64 : 0, narrow);
                    SW.bl = sw_best & 0xffff;
                    SW.bs = sw_best >> 16;
                    const bool done = fz_sweep_window<W>(SW, F, fz_sweep_factor(mode, v, F.la[0], lb0), cur_now() - 1e-6);
                    sw_win = SW.w | SW.w_end << 16;
                    sw_best = SW.bl | SW.bs << 16;
                    if (done) {
                        const double sc = fz_sweep_score(mode, v, F.la[0], lb0, fz_ratio_of(SW.bl, SW.bs));
                        best.take(sc, A.b_meta2[slot].w);
                        if (sc > 0.0) atomicMax(&s_best, (unsigned long long)__double_as_longlong(sc));
                        sw_item = -1;
                    }
                }
                if (A.phase_ticks && lane == 0) {
                    s_ticks[12] += n_busy;           // windows swept
                    s_ticks[13] += 64;               // ... and what the wave paid for
                }
            }
        };
        auto score_slot = [&](int entry, bool active, bool flush) {
            double sc = 0.0;
            int orig = -1, want_p = 0, slot_of = 0, sym_at = 0, sym_step = 0;
            int4 lens = make_int4(0, 0, 0, 0);
            bool did_score = false;
#ifdef PFZ_K7_PROFILE
            unsigned int sub[6] = {0, 0, 0, 0, 0, 0};
            const long long t_pop = clock64();
#endif
            if (active && A.exp != 1) {
                const int slot = entry & 0x7fffffff;
                const int4 m = A.b_meta[slot];
                orig = A.b_meta2[slot].w;
                FuzzTo T = to_of(slot, m);
                bool go = true;
                const double cur = cur_now();
                if (entry < 0) {
                    Meta x;
                    x.m = m;
                    x.m2 = A.b_meta2[slot];
                    x.m4 = make_int4(-1, -1, -1, -1);
                    x.h0 = A.b_hist[((slot >> 6) * 2 + 0) * 64 + (slot & 63)];
                    x.h1 = A.b_hist[((slot >> 6) * 2 + 1) * 64 + (slot & 63)];
                    const FuzzSummary sb = summary_of(x);
                    const int uu = fz_common_chars(sa, sb);
                    uint32_t ca, cb;
                    fz_intersect<W>(F, T, ca, cb);
                    const float ub = fz_upper_bound(sa, sb, mode, uu, ca ? 1 : 0, ca ? fz_token_set_bound<W>(F, ca, m.z, m.w, uu) : -1.0f);
                    go = !(ub + kBoundSlack < (float)cur);
                }
#ifdef PFZ_K7_PROFILE
                T.t0 = t_pop;
                for (int k = 0; k < 6; ++k) T.tk[k] = 0;
                FZ_TICK(T, 4);                          // the pair's metadata, the refined bound of a coarse entry
#endif
                if (go) {
                    sc = fz_score<W, true>(F, T, mode, cur, &want_p);
                    slot_of = slot;
                    lens = m;
                    sym_at = (int)(T.sym[0] - A.b_sym);
                    sym_step = (int)(T.sym[1] - T.sym[0]);
#ifdef PFZ_K7_PROFILE
                    FZ_TICK(T, 5);                      // (what follows the sweeps)
#endif
                    did_score = true;
                    best.take(sc, orig);
                    n_scored += 1;
                    n_steps += (unsigned int)work_estimate(F.la, m, mode, W);
                }
#ifdef PFZ_K7_PROFILE
                for (int k = 0; k < 6; ++k) sub[k] = T.tk[k];
#endif
            }
            if (A.phase_ticks) {
                // 8 batches, 9 lanes with a pair, 10 lanes scored, 11 lanes that swept windows, 12 windows swept,
                // 13 64 x the most windows any lane swept (what the wave paid for)
                const int n_act = __popcll(FZ_BALLOT(active)), n_go = __popcll(FZ_BALLOT(did_score)), n_sw = __popcll(FZ_BALLOT(want_p != 0));
#ifdef PFZ_K7_PROFILE
                // 16..21: the slowest lane's ticks by sub-phase -- tokens / set-up, LCS passes, token-set pass, window
                // sweeps, metadata + refined bound, the rest
                for (int k = 0; k < 6; ++k) {
                    unsigned int v = sub[k];
                    for (int d = 32; d >= 1; d >>= 1) v = max(v, (unsigned int)__shfl_xor((int)v, d, 64));
                    if (lane == 0) s_ticks[16 + k] += v;
                }
#endif
                if (lane == 0) {
                    s_ticks[8] += 1;
                    s_ticks[9] += n_act;
                    s_ticks[10] += n_go;
                    s_ticks[11] += n_sw;
                }
            }
            // publish the wave's best score to the workgroup
            double wb = active ? fmax(sc, 0.0) : 0.0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) wb = fmax(wb, __shfl_xor(wb, d, 64));
            if (lane == 0) {
                atomicMax(&s_best, (unsigned long long)__double_as_longlong(wb));
                if (cont_rec >= 0) {
                    // the units of a record tell each other: the best score any of them has found prunes for all
                    const unsigned long long mine = __hip_atomic_load(&s_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const unsigned long long theirs = atomicMax(&A.cont_cur[cont_rec], mine);
                    if (theirs > mine) atomicMax(&s_best, theirs);
                }
            }
            tick(3);
            // the sweeps this batch asks for: one item per run of windows (it-th item of the lane: forms in order)
            auto runs_of = [&](int bit, int la, int lb) {
                const int n_win = fz_n_windows(la, lb), sh = sweep_share_log2(n_win);
                return (want_p & bit) ? (n_win + (1 << sh) - 1) >> sh : 0;
            };
            const int runs0 = runs_of(1, F.la[0], lens.x), runs1 = runs_of(2, F.la[1], lens.y), runs2 = runs_of(4, F.la[2], lens.z);
            for (int it = 0;; ++it) {
                const bool has = it < runs0 + runs1 + runs2;
                const unsigned long long bal = FZ_BALLOT(has);
                if (has) {
                    const int v = it < runs0 ? 0 : (it < runs0 + runs1 ? 1 : 2);
                    const int run = it - (v == 0 ? 0 : (v == 1 ? runs0 : runs0 + runs1));
                    const int at = (sq_tail + lanes_below(bal)) & 63;
                    sweeps[0][at] = slot_of | v << 26 | run << 28;
                    sweeps[1][at] = (v == 0 ? lens.x : (v == 1 ? lens.y : lens.z)) | lens.x << 16;
                    sweeps[2][at] = sym_at + v * sym_step;
                }
                sq_tail += __popcll(bal);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                sweep_rounds(flush && bal == 0ull);
                if (bal == 0ull) break;
            }
            tick(6);
        };

        // ---- sweep 1: the bound of every pair of this unit's groups, left as a byte per pair in the workgroup's stretch of
        //      the bound cache (rounded up; bit 7: coarse), and every lane's best-bounded to-string: its seed.  Nothing but
        //      bounds in this loop.
        float seed_ub = -1.0f;
        int seed_slot = -1;
        const int row_region = is_cont ? __builtin_amdgcn_readfirstlane(A.cont_region[cont_rec]) : -1;      // (a continuation unit: its row's bytes)
        // (the region's base is wave-uniform -- a scalar register pair --, a pair's byte a 32-bit offset from it: n_groups * 64 <= 2^26)
        uint8_t *const ubc_row = A.ub_cache + (int64_t)(row_region >= 0 ? row_region : my_region) * A.n_groups * 64;
        auto ubc_at = [&](int g) -> uint8_t & { return ubc_row[(uint32_t)g * 64u + (uint32_t)lane]; };
        if (row_region < 0) {
            auto bound_group = [&](const Meta &x, int g) {
                bool valid, coarse;
                const float ub = bound_of(x, valid, coarse);
                n_bounded += 1;
                if (!is_cont && valid && !coarse && ub > seed_ub) {       // (a coarse bound says little: not a seed)
                    seed_ub = ub;
                    seed_slot = g * 64 + lane;
                }
                ubc_at(g) = valid ? (uint8_t)(min(127, (int)(fmaxf(ub, 0.0f) * 1.27f) + 1) | (coarse ? 128 : 0)) : (uint8_t)0;
            };
            // two groups per trip, each record fetched a group ahead into registers of its own (one record handed from trip
            // to trip cost seventeen register moves per group; the last trips re-read a group: harmless)
            Meta xa = load_meta(min(g_first, A.n_groups - 1));
            for (int g = g_first; g < A.n_groups; g += 2 * g_step) {
                const Meta xb = load_meta(min(g + g_step, A.n_groups - 1));
                bound_group(xa, g);
                xa = load_meta(min(g + 2 * g_step, A.n_groups - 1));
                if (g + g_step < A.n_groups) bound_group(xb, g + g_step);
            }
        }
        tick(1);
        // ---- sweep 2: the seeds are scored -- a real score to prune with (a continuation unit starts from its row's) --
        //      then the pairs whose byte reaches the best score so far, 64 at a time.  Two loops inside one: a tight one that
        //      walks the bytes until 64 pairs wait in the queue (nothing but a load, a compare and a push), and around it the
        //      ONE place where pairs are scored -- that code, by far the largest part of the kernel, is inlined exactly once;
        //      its last call empties the queue and finishes the window sweeps.
        int *queue = s_queue[wave];
        int q_head = 0, q_tail = 0;             // wave-uniform; entries [q_head, q_tail) of a ring of 128
        int batches = 0;
        bool swept_out = false;
        bool drain = !is_cont;                  // (the seeds: scored before anything else)
        if (drain) {
            const unsigned long long bal = FZ_BALLOT(seed_slot >= 0);
            if (seed_slot >= 0) queue[lanes_below(bal)] = seed_slot;
            q_tail = __popcll(bal);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        for (int g = g_first;;) {
            if (!drain && g < A.n_groups) {
                // (a byte b passes unless (float)b < thr: b >= ceil(thr), in integers -- and b != 0, the byte of a pair that is no candidate)
                const float thr = ((float)cur_now() - kBoundSlack) * 1.27f;       // (the best score moves only where pairs are scored)
                const int thr_i = __builtin_amdgcn_readfirstlane(max(1, (int)ceilf(fminf(thr, 1000.0f))));
                // (the walk's counters are wave-uniform; told so, the compiler keeps them -- and the loop's control -- in
                // scalar registers: the walk is a load, a compare and a push per group and was half bookkeeping)
                int gs = __builtin_amdgcn_readfirstlane(g), qt = __builtin_amdgcn_readfirstlane(q_tail);
                const int qh = __builtin_amdgcn_readfirstlane(q_head), gstep = __builtin_amdgcn_readfirstlane(g_step);
                // four groups' bytes per trip, the next trip's four already in flight (a group is no more than a compare and a
                // push: a load's latency per trip is all there is to this walk); when the queue fills up in between, the rest
                // are read again later
                int nx[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) nx[k] = ubc_at(min(gs + k * gstep, A.n_groups - 1));
                do {
                    int q4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) q4[k] = nx[k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) nx[k] = ubc_at(min(gs + (4 + k) * gstep, A.n_groups - 1));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (gs < A.n_groups && qt - qh < 64) {
                            const int q = q4[k], slot = gs * 64 + lane;
                            const bool want = (q & 127) >= thr_i && slot != seed_slot;
                            // (one ballot per compare: the ballot of a conjunction is rendered as a select and a compare more)
                            const unsigned long long bal = FZ_BALLOT((q & 127) >= thr_i) & FZ_BALLOT(slot != seed_slot);
                            if (want) queue[(qt + lanes_below(bal)) & 127] = slot | ((q & 128) ? (int)0x80000000 : 0);
                            qt += __popcll(bal);
                            n_bounded += 1;
                            gs += gstep;
                        }
                    }
                } while (gs < A.n_groups && qt - qh < 64);
                g = gs;
                q_tail = qt;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            tick(2);
            const bool last = !drain && g >= A.n_groups;
            // (a last call, with or without pairs, finishes the window sweeps under way)
            while (q_tail - q_head >= 64 || ((drain || last) && (q_tail > q_head || (last && !swept_out)))) {
                const bool active = lane < q_tail - q_head;
                const bool flush = (drain || last) && q_tail - q_head <= 64;
                score_slot(active ? queue[(q_head + lane) & 127] : -1, active, flush);
                swept_out = flush;
                q_head += min(64, q_tail - q_head);
                ++batches;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            if (last) break;
            if (!drain) continue;
            drain = false;
            // The seeds are scored: the pairs whose bound reaches the best of them are (at most) what is left to score -- one
            // quick pass over the bytes counts them.  A heavy row -- more than hand_batches batches of them -- is left to
            // several waves: as many continuation units as give each about half that (up to cont_parts), all of its groups
            // dealt among them, starting from the seeds' best
            if (!A.cont_list || A.n_groups - g <= A.hand_min_groups * g_step) continue;
            int survivors = 0;
            {
                const float thr = ((float)cur_now() - kBoundSlack) * 1.27f;
                for (int gg = g; gg < A.n_groups; gg += g_step) survivors += !((float)(ubc_at(gg) & 127) < thr) ? 1 : 0;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) survivors += __shfl_xor(survivors, d, 64);
            }
            // (a short from-string is heavy from fewer batches on: under WRatio and the partial scorers nearly every pair of
            // it sweeps windows -- the rows that END a launch are five-letter words with 50 batches)
            const int heavy = F.la[0] <= A.hand_short_len ? A.hand_short_batches : A.hand_batches;
            if (survivors > 64 * heavy) {
                const int units = min(A.cont_parts, max(2, (2 * survivors + 64 * heavy - 1) / (64 * heavy)));
                int at = 0;
                if (lane == 0) at = atomicAdd(A.n_cont, units);
                at = __builtin_amdgcn_readfirstlane(at);
                const bool fits = at + units <= A.cont_cap;
                if (fits) {
                    // the row's bytes go with it when there is a spare stretch for this workgroup to go on with (else the
                    // units bound their shares again)
                    int spare = 0;
                    if (lane == 0) spare = atomicAdd(A.region_next, 1);
                    spare = (int)gridDim.x + __builtin_amdgcn_readfirstlane(spare);
                    const bool moved = spare < A.n_regions;
                    if (lane == 0) {
                        __hip_atomic_store(&A.cont_region[at], moved ? my_region : -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&A.cont_cur[at], __hip_atomic_load(&s_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (moved) my_region = spare;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (lane < units && at + lane < A.cont_cap) {          // one record per unit (void ones when the list is full)
                    __hip_atomic_store(&A.cont_list[at + lane].x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&A.cont_list[at + lane].y, g + lane * g_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&A.cont_list[at + lane].z, g_step * units, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&A.cont_list[at + lane].w, fits ? (units | lane << 8) : -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (fits) g = A.n_groups;           // (the next trip finds nothing to walk: it finishes the seeds' window sweeps and ends the loop)
            }
        }

        // first best choice: (score desc, original index asc)
        wave_best(best);
        if (lane == 0) {
            red_s[wave] = best.score;
            red_i[wave] = best.idx;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < kK7Waves; ++w) best.take(red_s[w], red_i[w]);
            const int64_t o = (int64_t)A.row_slot[r] * A.n_parts_total + (is_cont ? A.cont_part0 : A.part0) + part;
            A.part_score[o] = best.score;
            A.part_idx[o] = best.idx;
            // (after a possible hand-over of this row: when every row is counted, the list of remainders is final)
            if (!is_cont && A.cont_list) __hip_atomic_fetch_add(A.rows_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (A.row_stats) {
            unsigned int ns = n_scored;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) ns += __shfl_xor(ns, d, 64);
            if (lane == 0) atomicAdd(&A.row_stats[2 * (int64_t)A.row_slot[r]], (unsigned long long)ns);
            if (tid == 0) {
                const long long t_end = wall_clock64();
                atomicAdd(&A.row_stats[2 * (int64_t)A.row_slot[r] + 1], (unsigned long long)(t_end - t_begin));
                // (after the timers: when the row's first unit began -- kept as 2^62 minus the time --, when its last ended)
                atomicMax(&A.phase_ticks[24 + 2 * (int64_t)A.row_slot[r]], (1ull << 62) - (unsigned long long)t_begin);
                atomicMax(&A.phase_ticks[24 + 2 * (int64_t)A.row_slot[r] + 1], (unsigned long long)t_end);
            }
        }
        if (A.counters) {
            unsigned long long nb = n_bounded, nsc = n_scored, nst = n_steps;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                nb += __shfl_xor(nb, d, 64);
                nsc += __shfl_xor(nsc, d, 64);
                nst += __shfl_xor(nst, d, 64);
            }
            if (lane == 0) {
                atomicAdd(&A.counters[0], nb);
                atomicAdd(&A.counters[1], nsc);
                atomicAdd(&A.counters[2], nst);
            }
        }
        // clear this from-string's table entries
        for (int v = 0; v < 3; ++v) {
            const int m = s_la[v];
            for (int p = tid; p < m; p += kK7Threads) {
                const uint32_t c = load_unit(A.a_form[v], A.a_width, a0 + p);
                const int sy = c < A.lut_len ? (int)A.lut[c] : 0;
#pragma unroll
                for (int w = 0; w < W; ++w) pm[(sy * 3 + v) * W + w] = 0ull;
            }
        }
        __syncthreads();
    }
    if (A.phase_ticks) {
        __syncthreads();
        if (tid < 24) atomicAdd(&A.phase_ticks[tid], s_ticks[tid]);
    }
}

// the best of a from-string over all the parts every launch left (same order: score desc, original index asc)
__global__ __launch_bounds__(256) void k7_merge_parts(const double *__restrict__ part_score, const int32_t *__restrict__ part_idx,
                                                       int64_t n_rows, int32_t n_parts, int32_t *__restrict__ out_idx,
                                                       double *__restrict__ out_score)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    RowBest b = {-1.0, INT_MAX};
    for (int p = 0; p < n_parts; ++p) b.take(part_score[r * n_parts + p], part_idx[r * n_parts + p]);
    out_idx[r] = b.idx == INT_MAX ? -1 : b.idx;
    out_score[r] = b.idx == INT_MAX ? 0.0 : b.score;
}

// (index, float64 score) -> the two-column result buffer the sharded jobs all-gather: idx[r][0] = index, the score's bits in
// the two value lanes of the row
__global__ __launch_bounds__(256) void k_best_to_topn(const int32_t *__restrict__ idx, const double *__restrict__ score, int64_t n,
                                                       int32_t *__restrict__ t_idx, float *__restrict__ t_val)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    t_idx[2 * r] = idx[r];
    t_idx[2 * r + 1] = -1;
    ((double *)t_val)[r] = score[r];
}

struct DevBuf {
    pfz_ctx *ctx = nullptr;
    void *p = nullptr;
    explicit DevBuf(pfz_ctx *c) : ctx(c) {}
    ~DevBuf() { if (p) pool_free(p); }
    int alloc(size_t bytes) { return pool_alloc(ctx, &p, bytes > 0 ? bytes : 16); }
    template <typename T> int upload(const std::vector<T> &v)
    {
        PFZ_TRY(alloc(v.size() * sizeof(T)));
        if (!v.empty()) PFZ_TRY(copy_h2d(ctx, p, v.data(), v.size() * sizeof(T)));
        return PFZ_OK;
    }
};

int fuzz_general_launch(pfz_ctx *ctx, FuzzArgs A, const pfz_strings *F, const pfz_strings *T, const std::vector<int32_t> &rows);   // k7_general.hip

// rows [begin, end) of `from` against all of `to`: (first best index, score) into device buffers d_idx / d_score of end - begin entries
static int fuzz_run(pfz_ctx *ctx, const pfz_strings *F_c, const pfz_strings *T_c, int32_t scorer, const int32_t *skip_idx, int64_t begin,
                    int64_t end, int32_t *d_idx, double *d_score, unsigned long long *h_counters)
{
    pfz_strings *F = const_cast<pfz_strings *>(F_c), *T = const_cast<pfz_strings *>(T_c);     // the caches live inside the handles
    const int64_t n_rows = end - begin, n_to = T->n;
    if (T->n >= INT_MAX - 64 || F->n >= INT_MAX) {
        set_error("pfz_fuzz: list too long");
        return PFZ_ERR_UNSUPPORTED;
    }
    PFZ_TRY(ensure_forms(ctx, F));
    if (!T->fuzz_plan) PFZ_TRY(build_plan(ctx, T));
    const pfz_fuzz_plan *pl = T->fuzz_plan;
    const pfz_fuzz_forms *ff = F->fuzz_forms;

    // token ids of the from-list in the to-list's table (the same handle: the plan's own)
    DevBuf d_aid(ctx), d_skip(ctx), d_counters(ctx);
    const int32_t *a_tok_id = pl->t_tok_id;
    if (F != T) {
        PFZ_TRY(d_aid.alloc((size_t)ff->tok_cap * sizeof(int32_t)));
        if (F->n > 0) {
            ProfScope ps(ctx, "k7_prepare");
            PFZ_TRY(from_token_ids(ctx, F, T, (int32_t *)d_aid.p));
        }
        a_tok_id = (const int32_t *)d_aid.p;
    }
    int skip_up_to = 0;
    if (skip_idx) {
        std::vector<int32_t> codes(skip_idx, skip_idx + F->n);
        skip_up_to = decode_skip_codes(codes);
        PFZ_REQUIRE(skip_up_to >= 0, "pfz_fuzz_extract_one: skip_idx mixes single choices (>= 0) and 'up to' codes (<= -2)");
        PFZ_TRY(d_skip.alloc((size_t)F->n * sizeof(int32_t)));
        PFZ_TRY(copy_h2d(ctx, d_skip.p, codes.data(), (size_t)F->n * sizeof(int32_t)));
    }
    if (h_counters) {
        PFZ_TRY(d_counters.alloc(4 * sizeof(unsigned long long)));
        PFZ_HIP(hipMemsetAsync(d_counters.p, 0, 4 * sizeof(unsigned long long), ctx->stream));
    }

    // word classes of the from-strings; beyond 256 characters / 32 distinct tokens / a 60 KiB match table: the general kernel
    static const int kWords[3] = {1, 2, 4};
    std::vector<int32_t> cls[4], slot_of[4];
    const bool force_general = getenv("PFZ_K7_FORCE_GENERAL") != nullptr;
    for (int64_t i = begin; i < end; ++i) {
        const int64_t len = F->h_off[(size_t)i + 1] - F->h_off[(size_t)i];
        int c = len > 256 ? 3 : (len > 128 ? 2 : (len > 64 ? 1 : 0));
        if (ff->h_ntok[(size_t)i] > kFuzzMaxTokens || force_general) c = 3;
        if (c < 3 && (size_t)(pl->n_sym + 1) * 3 * kWords[c] * sizeof(uint64_t) > 60 * 1024) c = 3;
        cls[c].push_back((int32_t)i);
        slot_of[c].push_back((int32_t)(i - begin));
    }
    // the units are handed out in this order: the long from-strings -- many more of their pairs survive the bound -- first,
    // so that none of them starts when the others are about to finish (a counting sort by length: O(n)).  (Round 5 tried the
    // short strings -- the next most expensive per row under the window-sweeping scorers -- right behind the long ones: 7.91 ms
    // against 7.56 on one box.  What a launch ends on are the continuation units of heavy rows of EVERY length, dealt after
    // the last primary unit; handing rows over earlier -- 32 / 48 / 24 / 16 batches instead of 64 -- costs more than it evens
    // out: 8.33 / 7.90 / 9.83 / 11.79 ms against 7.73.)
    for (int c = 0; c < 3; ++c) {
        const size_t n = cls[c].size();
        if (n < 2) continue;
        std::vector<int32_t> start(258, 0), rows(n), slots(n);
        auto len_of = [&](int32_t i) { return (int)(F->h_off[(size_t)i + 1] - F->h_off[(size_t)i]); };
        for (size_t k = 0; k < n; ++k) start[(size_t)(256 - len_of(cls[c][k])) + 1]++;
        for (size_t l = 1; l < start.size(); ++l) start[l] += start[l - 1];
        for (size_t k = 0; k < n; ++k) {
            const size_t at = (size_t)start[(size_t)(256 - len_of(cls[c][k]))]++;
            rows[at] = cls[c][k];
            slots[at] = slot_of[c][k];
        }
        cls[c].swap(rows);
        slot_of[c].swap(slots);
    }

    FuzzArgs A{};
    A.a_form[0] = F->chars;
    A.a_form[1] = ff->form1;
    A.a_form[2] = ff->form2;
    A.a_width = F->char_width;
    A.a_off = F->offsets;
    A.a_len1 = ff->len1;
    A.a_len2 = ff->len2;
    A.a_ntok = ff->ntok;
    A.a_ntok_all = ff->ntok_all;
    A.a_tok_len = ff->tok_len;
    A.a_tok_id = a_tok_id;
    A.lut = pl->lut;
    A.lut_len = pl->lut_len;
    A.cls = pl->cls;
    A.space_rank = pl->space_rank;
    A.space_class = pl->space_class;
    A.b_sym = pl->sym;
    A.b_tag = pl->tag;
    A.b_tok_id = pl->tok_id;
    A.b_tok_len = pl->tok_len;
    A.b_meta = pl->meta;
    A.b_meta2 = pl->meta2;
    A.b_meta3 = pl->meta3;
    A.b_meta4 = pl->meta4;
    A.b_hist = pl->hist;
    A.b_pres = pl->pres;
    A.n_groups = (int32_t)pl->n_groups;
    A.n_sym1 = pl->n_sym + 1;
    A.mode = scorer;
    A.skip_idx = skip_idx ? (const int32_t *)d_skip.p : nullptr;
    A.skip_up_to = skip_up_to;
    A.counters = h_counters ? (unsigned long long *)d_counters.p : nullptr;

    // how many workgroups share a from-string's to-groups (few from-strings: split, as K4 does), per class; then one slot for
    // the general kernel's "to-strings with more than 32 tokens" pass and one for its own rows
    const int64_t max_grid = (int64_t)ctx->prop.multiProcessorCount * (4 * PFZ_K7_OCC / kK7Waves);      // (persistent workgroups: units are handed out)
    int32_t parts_of[3] = {1, 1, 1}, max_parts = 1;
    for (int c = 0; c < 3; ++c) {
        if (cls[c].empty()) continue;
        const int64_t n = (int64_t)cls[c].size();
        const int64_t want = (2 * max_grid + n - 1) / n, cap = std::max<int64_t>(1, pl->n_groups / (4 * kK7Waves));
        parts_of[c] = (int32_t)std::max<int64_t>(1, std::min(want, cap));
        // One-word from-strings that are ONE unit each hand their heavy rows over to continuation units (below) -- rows
        // that are split up front cannot, and a few thousand of them split in two are the worst of both: 5 000 x 20 000
        // titles, WRatio, 9.3 ms in two parts each against 5.2 ms whole (2 500: 5.2 against 4.6; 1 000 rows are better
        // split: 2.3 against 3.2).  From 2 048 rows on a row is one unit.
        if (c == 0 && n >= 2048 && kK7Waves == 1) parts_of[c] = 1;
        if (const char *e = getenv("PFZ_K7_PARTS")) parts_of[c] = std::max(1, atoi(e));
        max_parts = std::max(max_parts, parts_of[c]);
    }
    int32_t hand_batches = kHandBatches, hand_min_groups = kHandMinGroups, cont_parts = kContParts;
    int32_t short_len = kHandShortLen, short_batches = kHandShortBatches;
    if (const char *e = getenv("PFZ_K7_HAND_BATCHES")) hand_batches = atoi(e), hand_min_groups = 0;      // tests: hand over early
    cont_parts = std::max(1, std::min(cont_parts, 64));
    const int32_t n_parts_total = max_parts + 1 + cont_parts;
    DevBuf d_ps(ctx), d_pi(ctx);
    PFZ_TRY(d_ps.alloc((size_t)n_rows * n_parts_total * sizeof(double)));
    PFZ_TRY(d_pi.alloc((size_t)n_rows * n_parts_total * sizeof(int32_t)));
    PFZ_HIP(hipMemsetAsync(d_ps.p, 0xff, (size_t)n_rows * n_parts_total * sizeof(double), ctx->stream));      // NaN bits: never taken ...
    PFZ_HIP(hipMemsetAsync(d_pi.p, 0x7f, (size_t)n_rows * n_parts_total * sizeof(int32_t), ctx->stream));     // ... and a huge index
    A.n_parts_total = n_parts_total;
    A.part_score = (double *)d_ps.p;
    A.part_idx = (int32_t *)d_pi.p;

    DevBuf d_rows[4] = {DevBuf(ctx), DevBuf(ctx), DevBuf(ctx), DevBuf(ctx)}, d_slots[4] = {DevBuf(ctx), DevBuf(ctx), DevBuf(ctx), DevBuf(ctx)};
    DevBuf d_next(ctx), d_stats(ctx), d_cont(ctx), d_cont_cur(ctx);
    PFZ_TRY(d_next.alloc(16 * sizeof(int32_t)));          // [0..2] unit counters, [4..6] continuation unit counters, [8..10] continuation counts
    PFZ_HIP(hipMemsetAsync(d_next.p, 0, 16 * sizeof(int32_t), ctx->stream));
    const int32_t cont_cap = (int32_t)std::min<int64_t>(4 * n_rows + 64, 1 << 22);      // (continuation units of one launch, at most)
    PFZ_TRY(d_cont.alloc((size_t)cont_cap * 3 * sizeof(int4)));
    PFZ_TRY(d_cont_cur.alloc((size_t)cont_cap * 3 * sizeof(unsigned long long)));
    // the bound cache: a byte per (workgroup, to-slot), a stretch of its own for every launch (they may run side by side).
    // Few enough persistent workgroups that it stays within a budget: 1/16 of the device's memory, 16 GiB at most (288 GB:
    // 4096 workgroups up to four million to-strings) -- of the TOTAL, not of what happens to be free, so that the sizes,
    // and with them the blocks the caching allocator holds, are the same on every call
    const int64_t ubc_budget = std::max<int64_t>((int64_t)64 << 20, std::min<int64_t>((int64_t)16 << 30, (int64_t)(ctx->prop.totalGlobalMem / 16)));
    const int64_t grid_cap = std::max<int64_t>(64, std::min<int64_t>(max_grid, ubc_budget / (3 * std::max<int64_t>(pl->n_groups, 1) * 64)));
    auto grid_of = [&](int c, bool hand) { return hand ? grid_cap : std::min<int64_t>((int64_t)cls[c].size() * parts_of[c], grid_cap); };
    const bool hand_over = !getenv("PFZ_K7_NO_HANDOVER");
    // (a launch that hands rows over has spare stretches: a handed-over row keeps the one its bytes are in; when they run out,
    // the continuation units of the rows handed over after that bound their shares again)
    DevBuf d_ubc(ctx), d_cont_region(ctx);
    int64_t ubc_at[3] = {0, 0, 0}, ubc_regions[3] = {0, 0, 0}, ubc_slots = 0;
    const int64_t region_bytes = std::max<int64_t>(pl->n_groups, 1) * 64;
    for (int c = 0; c < 3; ++c) {
        ubc_at[c] = ubc_slots;
        if (cls[c].empty()) continue;
        const bool hand = hand_over && parts_of[c] == 1 && kK7Waves == 1;
        // (half the rows at most are ever heavy; an eighth of the budget -- 2 GiB -- of spare stretches at most)
        const int64_t spare = hand ? std::min<int64_t>((int64_t)cls[c].size() / 2 + 64, (ubc_budget / 8) / region_bytes) : 0;
        ubc_regions[c] = grid_of(c, hand) + spare;
        ubc_slots += ubc_regions[c] * region_bytes;
    }
    PFZ_TRY(d_ubc.alloc((size_t)std::max<int64_t>(ubc_slots, 64)));
    PFZ_TRY(d_cont_region.alloc((size_t)cont_cap * 3 * sizeof(int32_t)));
    const char *stats_path = getenv("PFZ_K7_ROW_STATS");
    if (stats_path) {
        // (two per row, then eight phase timers: see the kernel)
        PFZ_TRY(d_stats.alloc(((size_t)n_rows * 4 + 24) * sizeof(unsigned long long)));
        PFZ_HIP(hipMemsetAsync(d_stats.p, 0, ((size_t)n_rows * 4 + 24) * sizeof(unsigned long long), ctx->stream));
        A.row_stats = (unsigned long long *)d_stats.p;
        A.phase_ticks = A.row_stats + (size_t)n_rows * 2;
    }
#ifdef PFZ_EXPERIMENTS      // (variant builds only, tools/build_variant.sh -DPFZ_EXPERIMENTS: results wrong on purpose)
    if (const char *e = getenv("PFZ_K7_EXP")) A.exp = atoi(e);
#endif
    const bool side = !cls[0].empty() && (!cls[1].empty() || !cls[2].empty()) && !getenv("PFZ_K7_NO_SIDE_STREAM");
    bool used_side = false;
    if (side) PFZ_TRY(ensure_side_stream(ctx));
    // an error return between a side-stream launch and its join must not free the buffers that launch is still using
    // (the DevBufs above are destroyed after this guard: it is declared after them)
    struct SideGuard {
        pfz_ctx *ctx;
        bool *used;
        ~SideGuard() { if (*used) (void)hipStreamSynchronize(ctx->stream2); }
    } side_guard{ctx, &used_side};
    ProfScope ps_all(ctx, "k7_fuzz");
    for (int c = 2; c >= 0; --c) {       // (the side-stream launches first: the persistent waves of class 0 would keep them out)
        if (cls[c].empty() || n_to == 0) continue;
        PFZ_TRY(d_rows[c].upload(cls[c]));
        PFZ_TRY(d_slots[c].upload(slot_of[c]));
        A.rows = (const int32_t *)d_rows[c].p;
        A.row_slot = (const int32_t *)d_slots[c].p;
        A.n_rows = (int32_t)cls[c].size();
        A.parts = parts_of[c];
        A.part0 = 0;
        // (the match table, the scratch columns -- bytes for a small alphabet --, the timers when asked for: see the kernel)
        const size_t lds = (size_t)A.n_sym1 * 3 * (size_t)kWords[c] * sizeof(uint64_t) +
                           (size_t)kK7Waves * kFuzzStage * 64 * (A.n_sym1 <= 256 ? 1 : 2) + (A.phase_ticks ? 24 * sizeof(unsigned long long) : 0);
        // (only where a row is ONE unit: a row already split over several units would hand over several remainders, and they
        // would share the continuation's result slots)
        const bool hand = hand_over && A.parts == 1 && kK7Waves == 1;
        A.cont_list = hand ? (int4 *)d_cont.p + (size_t)c * cont_cap : nullptr;
        A.cont_cur = (unsigned long long *)d_cont_cur.p + (size_t)c * cont_cap;
        A.n_cont = (int32_t *)d_next.p + 8 + c;
        A.cont_cap = cont_cap;
        A.cont_parts = cont_parts;
        A.hand_batches = std::max(1, hand_batches);
        A.hand_short_len = mode_sweeps_windows(scorer) ? short_len : 0;
        A.hand_short_batches = std::max(1, std::min(short_batches, A.hand_batches));
        A.hand_min_groups = hand_min_groups;
        A.cont_part0 = max_parts + 1;
        A.rows_done = (int32_t *)d_next.p + 4 + c;
        A.next_unit = (int32_t *)d_next.p + c;
        if (hand) PFZ_HIP(hipMemsetAsync(A.cont_list, 0, (size_t)cont_cap * sizeof(int4), ctx->stream));
        // the classes of long from-strings (few rows, every one split over many units) run beside the first on the side stream
        hipStream_t st = ctx->stream;
        if (c > 0 && side) {
            PFZ_HIP(hipEventRecord(ctx->side_events[0], ctx->stream));      // (inputs, counters and the cleared lists are ready)
            PFZ_HIP(hipStreamWaitEvent(ctx->stream2, ctx->side_events[0], 0));
            st = ctx->stream2;
            used_side = true;
        }
        FuzzArgs L = A;
        L.ub_cache = (uint8_t *)d_ubc.p + ubc_at[c];
        L.n_regions = (int32_t)std::min<int64_t>(ubc_regions[c], INT32_MAX);
        L.region_next = (int32_t *)d_next.p + 12 + c;
        L.cont_region = (int32_t *)d_cont_region.p + (size_t)c * cont_cap;
        // persistent one-wave workgroups: the rows, then -- in the same launch -- the remainders of the heavy ones
        const unsigned grid = (unsigned)grid_of(c, hand);
        // WRatio -- what PolyFuzz("EditDistance") and RapidFuzz() run by default -- has instances of its own: the scorer a
        // compile-time constant (no scalar dispatch inside the sweeps, only that scorer's values live)
        if (scorer == kWRatio) {
            if (c == 0) hipLaunchKernelGGL((k7_fuzz_kernel<1, kWRatio>), dim3(grid), dim3(kK7Threads), lds, st, L);
            else if (c == 1) hipLaunchKernelGGL((k7_fuzz_kernel<2, kWRatio>), dim3(grid), dim3(kK7Threads), lds, st, L);
            else hipLaunchKernelGGL((k7_fuzz_kernel<4, kWRatio>), dim3(grid), dim3(kK7Threads), lds, st, L);
        } else if (c == 0) hipLaunchKernelGGL(k7_fuzz_kernel<1>, dim3(grid), dim3(kK7Threads), lds, st, L);
        else if (c == 1) hipLaunchKernelGGL(k7_fuzz_kernel<2>, dim3(grid), dim3(kK7Threads), lds, st, L);
        else hipLaunchKernelGGL(k7_fuzz_kernel<4>, dim3(grid), dim3(kK7Threads), lds, st, L);
        PFZ_HIP(hipGetLastError());
    }
    if (used_side) {
        PFZ_HIP(hipEventRecord(ctx->side_events[1], ctx->stream2));
        PFZ_HIP(hipStreamWaitEvent(ctx->stream, ctx->side_events[1], 0));
        used_side = false;          // joined: from here on the main stream orders everything (the guard stands down)
    }
    A.cont_list = nullptr;
    if (n_to > 0) {
        // the general kernel: its own from-rows against every to-string, and every OTHER from-row against the to-strings
        // the kernels above left out (more than 32 distinct tokens)
        A.parts = 1;
        A.part0 = max_parts;
        if (!cls[3].empty()) {
            PFZ_TRY(d_slots[3].upload(slot_of[3]));
            A.row_slot = (const int32_t *)d_slots[3].p;
            A.big_slots = nullptr;
            A.n_big = 0;
            PFZ_TRY(fuzz_general_launch(ctx, A, F, T, cls[3]));
        }
        if (!pl->big_slots.empty()) {
            std::vector<int32_t> others, other_slots;
            for (int c = 0; c < 3; ++c) {
                others.insert(others.end(), cls[c].begin(), cls[c].end());
                other_slots.insert(other_slots.end(), slot_of[c].begin(), slot_of[c].end());
            }
            if (!others.empty()) {
                DevBuf d_os(ctx);
                PFZ_TRY(d_os.upload(other_slots));
                A.row_slot = (const int32_t *)d_os.p;
                A.big_slots = pl->d_big_slots;
                A.n_big = (int32_t)pl->big_slots.size();
                    PFZ_TRY(fuzz_general_launch(ctx, A, F, T, others));
            }
        }
    }
    {
        hipLaunchKernelGGL(k7_merge_parts, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)d_ps.p,
                           (const int32_t *)d_pi.p, n_rows, n_parts_total, d_idx, d_score);
        PFZ_HIP(hipGetLastError());
    }
    if (h_counters) PFZ_TRY(copy_d2h(ctx, h_counters, d_counters.p, 4 * sizeof(unsigned long long)));
    if (stats_path) {
        std::vector<unsigned long long> st((size_t)n_rows * 4 + 24);
        PFZ_TRY(copy_d2h(ctx, st.data(), d_stats.p, st.size() * sizeof(unsigned long long)));
        if (FILE *fp = fopen(stats_path, "wb")) {
            fwrite(st.data(), sizeof(unsigned long long), st.size(), fp);
            fclose(fp);
        }
    }
    return PFZ_OK;
}

const int32_t *fuzz_forms_tok_pos(const pfz_strings *S) { return S->fuzz_forms ? S->fuzz_forms->tok_pos : nullptr; }

int best_to_topn(pfz_ctx *ctx, const int32_t *d_idx, const double *d_score, int64_t n, pfz_topn *out)
{
    if (n == 0) return PFZ_OK;
    hipLaunchKernelGGL(k_best_to_topn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_idx, d_score, n, out->idx, out->val);
    PFZ_HIP(hipGetLastError());
    return PFZ_OK;
}

static int check_args(pfz_ctx *ctx, const pfz_strings *F, const pfz_strings *T, int32_t scorer, int64_t begin, int64_t end, const char *who)
{
    PFZ_REQUIRE(ctx && F && T, "%s: NULL argument", who);
    PFZ_REQUIRE(scorer >= kWRatio && scorer <= kPartialTokenRatio, "%s: unknown scorer %d", who, scorer);
    PFZ_REQUIRE(begin >= 0 && begin <= end && end <= F->n, "%s: row range [%lld,%lld) outside [0,%lld)", who, (long long)begin,
                (long long)end, (long long)F->n);
    return PFZ_OK;
}

}  // namespace pfz

using namespace pfz;

extern "C" {

int pfz_fuzz_extract_one(pfz_ctx *ctx, const pfz_strings *from_strings, const pfz_strings *to_strings, int32_t scorer,
                         const int32_t *skip_idx, int64_t from_begin, int64_t from_end, int32_t *out_idx, double *out_score)
{
    PFZ_TRY(check_args(ctx, from_strings, to_strings, scorer, from_begin, from_end, "pfz_fuzz_extract_one"));
    PFZ_REQUIRE(out_idx && out_score, "pfz_fuzz_extract_one: NULL output");
    const int64_t n = from_end - from_begin;
    if (n == 0) return PFZ_OK;
    PFZ_HIP(hipSetDevice(ctx->device));
    DevBuf d_idx(ctx), d_score(ctx);
    PFZ_TRY(d_idx.alloc((size_t)n * sizeof(int32_t)));
    PFZ_TRY(d_score.alloc((size_t)n * sizeof(double)));
    PFZ_TRY(fuzz_run(ctx, from_strings, to_strings, scorer, skip_idx, from_begin, from_end, (int32_t *)d_idx.p, (double *)d_score.p, nullptr));
    PFZ_TRY(copy_d2h(ctx, out_idx, d_idx.p, (size_t)n * sizeof(int32_t)));
    PFZ_TRY(copy_d2h(ctx, out_score, d_score.p, (size_t)n * sizeof(double)));
    PFZ_HIP(hipStreamSynchronize(ctx->stream));
    return PFZ_OK;
}

int pfz_fuzz_extract_one_dev(pfz_ctx *ctx, const pfz_strings *from_strings, const pfz_strings *to_strings, int32_t scorer,
                             const int32_t *skip_idx, int64_t from_begin, int64_t from_end, pfz_topn *out, uint64_t *work_counters)
{
    PFZ_TRY(check_args(ctx, from_strings, to_strings, scorer, from_begin, from_end, "pfz_fuzz_extract_one_dev"));
    const int64_t n = from_end - from_begin;
    PFZ_REQUIRE(out && out->ntop == 2 && out->n_rows >= n, "pfz_fuzz_extract_one_dev: the result buffer must have 2 columns and >= %lld rows",
                (long long)n);
    if (work_counters) work_counters[0] = work_counters[1] = work_counters[2] = work_counters[3] = 0;
    if (n == 0) return PFZ_OK;
    PFZ_HIP(hipSetDevice(ctx->device));
    DevBuf d_idx(ctx), d_score(ctx);
    PFZ_TRY(d_idx.alloc((size_t)n * sizeof(int32_t)));
    PFZ_TRY(d_score.alloc((size_t)n * sizeof(double)));
    PFZ_TRY(fuzz_run(ctx, from_strings, to_strings, scorer, skip_idx, from_begin, from_end, (int32_t *)d_idx.p, (double *)d_score.p,
                     (unsigned long long *)work_counters));
    return best_to_topn(ctx, (const int32_t *)d_idx.p, (const double *)d_score.p, n, out);
}

int pfz_fuzz_plan_info(pfz_ctx *ctx, const pfz_strings *to_strings, int64_t *n_symbols, int64_t *n_groups, int64_t *n_tokens,
                       int64_t *n_general_strings)
{
    PFZ_REQUIRE(ctx && to_strings, "pfz_fuzz_plan_info: NULL argument");
    pfz_strings *T = const_cast<pfz_strings *>(to_strings);
    if (!T->fuzz_plan) {
        PFZ_HIP(hipSetDevice(ctx->device));
        PFZ_TRY(build_plan(ctx, T));
    }
    if (n_symbols) *n_symbols = T->fuzz_plan->n_sym;
    if (n_groups) *n_groups = T->fuzz_plan->n_groups;
    if (n_tokens) {
        *n_tokens = 0;
        for (int32_t t : T->fuzz_forms->h_ntok) *n_tokens += t;
    }
    if (n_general_strings) *n_general_strings = (int64_t)T->fuzz_plan->big_slots.size();
    return PFZ_OK;
}

}  // extern "C"
