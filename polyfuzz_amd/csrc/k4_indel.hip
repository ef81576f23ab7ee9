// K4 -- all-pairs Indel ratio (rapidfuzz.fuzz.ratio) with fused row arg-max.
//
// Replaces the hot loop of EditDistance._calculate_edit_distance, reference
// polyfuzz/models/_distance.py:89-102: scorer(from, to) for every to-string,
// np.argmax (first maximum), np.max.  fuzz.ratio = (1 - (|a|+|b|-2*LCS(a,b)) /
// (|a|+|b|)) * 100 on code points, float64.
//
// Algorithm: bit-parallel LCS (Crochemore/Hyyro): with PM[c] = positions of
// character c in the from-string and V = all ones, every to-character does
//     u = V & PM[c];  V = (V + u) | (V ^ u)
// and LCS = number of zero bits of V.  It is integer work, exact, and needs
// ~5 VALU operations per to-character per 32/64-bit word instead of |a| DP
// cells -- the anti-diagonal wavefront the DP suggests would do 16-64x more work.
//
// Mapping to CDNA4
//   workgroup (4 waves) = one from-string at a time: its PM table lives in LDS
//     (one word per alphabet symbol; symbols are ranks of the distinct code
//     points of the TO-list, so any Unicode input works -- a from-character the
//     to-list never uses is never looked up and gets no entry), built with ds_or
//     and cleared by re-visiting the from-string's own characters.
//   lane = one to-string.  To-strings are sorted by length and stored in groups
//     of 64 as [t/4][lane] dwords of 4 packed symbols (2 for >255 symbols), so a
//     wave reads 256 contiguous bytes per step and lanes of a group finish
//     together; padding symbol 0 has an empty PM entry and is a no-op.
//   from-strings are bucketed by length into word classes (32-bit word for <= 32
//     characters, 1..16 64-bit words beyond) so V stays in registers.
//   arg-max: per lane (score desc, original index asc) in float64 with the exact
//     reference formula, then wave shuffles + one LDS step per from-string.
// The to-side "plan" -- alphabet, length-sorted groups, packed symbols -- depends on
// the to-list alone; it is built once per to-list (packing on the device) and cached
// on its pfz_strings handle, so a repeated to-list costs no preparation at all.
// Roofline: integer VALU + LDS lookups; HBM traffic is the to-strings once per
// from-string out of L2 (0.3 MB) -- not HBM-bound.
#include "pfz_internal.h"

#include <algorithm>
#include <limits.h>
#include <string.h>

#ifndef PFZ_K4_EXP
#define PFZ_K4_EXP 0
#endif

namespace pfz {

// s | (v & ~u) -- = s | (v ^ u) where u is a subset of v -- as ONE v_bitop3_b32 per 32 bits (truth table 0xF4): the kernels of
// this file are bound by their vector instruction rate (SQ counters: 85 % of the issue slots), and the compiler renders the
// xor and the or as two instructions
__device__ inline uint32_t or_andnot(uint32_t s, uint32_t v, uint32_t u) { return __builtin_amdgcn_bitop3_b32(s, v, u, 0xF4); }
__device__ inline uint64_t or_andnot(uint64_t s, uint64_t v, uint64_t u)
{
    return (uint64_t)or_andnot((uint32_t)(s >> 32), (uint32_t)(v >> 32), (uint32_t)(u >> 32)) << 32 | or_andnot((uint32_t)s, (uint32_t)v, (uint32_t)u);
}

template <typename WORD, int W>
__device__ inline void lcs_step(WORD (&V)[W], const WORD *__restrict__ pmc)
{
    WORD carry = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const WORD M = pmc[w];
        const WORD s = V[w];
        const WORD u = s & M;
        WORD sum = s + u;
        const WORD c1 = sum < s ? 1 : 0;
        sum += carry;
        const WORD c2 = sum < carry ? 1 : 0;
        carry = c1 | c2;
        V[w] = or_andnot(sum, s, u);
    }
}

__device__ inline void lds_or(uint32_t *p, uint32_t v) { atomicOr(p, v); }
__device__ inline void lds_or(uint64_t *p, uint64_t v) { atomicOr((unsigned long long *)p, (unsigned long long)v); }
__device__ inline int popc_word(uint32_t v) { return __popc(v); }
__device__ inline int popc_word(uint64_t v) { return __popcll(v); }

struct IndelArgs {
    const void *a_chars;       // from-strings: code units of a_width bytes
    int32_t a_width;
    const int64_t *a_off;      // [n_from + 1]
    const uint16_t *lut;       // code unit -> symbol rank (0 = not in the to-list's alphabet), lut_len entries
    uint32_t lut_len;
    const int32_t *rows;       // from-rows of this word class
    int32_t n_rows;
    const uint32_t *b_packed;  // to-strings, groups of 64, [t/PER][lane]
    const int64_t *g_off;      // [n_groups] dword offset of each group
    const int32_t *g_steps;    // [n_groups] dwords per lane
    const int32_t *b_len;      // [n_groups*64]
    const int32_t *b_orig;     // [n_groups*64] original to-index, -1 = padding lane
    int32_t n_groups;
    const int32_t *skip_idx;   // [n_from] or NULL (decoded: pfz_internal.h decode_skip_codes)
    int32_t skip_up_to;        // 0: choice skip_idx[row] is left out; 1: every choice up to skip_idx[row] is
    int32_t n_sym1;            // alphabet size + 1 (symbol 0 = padding)
    int64_t from_begin;
    int64_t n_to;
    int32_t *out_idx;          // [from_end - from_begin]
    double *out_score;
    double *matrix;            // optional [(from_end-from_begin) * n_to]
    int32_t parts;             // the to-groups of a from-string (or quad) are split over `parts` workgroups ...
    int32_t *partial;          // ... which leave (lcs, maximum, idx) per (class row, part) here for k4_merge_parts
};

// The running best of a from-string is kept as the exact rational lcs / maximum (maximum = |a| + |b|): the ratio
// (1 - (maximum - 2 lcs) / maximum) * 100 is monotone in it, two different rationals with denominators < 2^32 differ
// by far more than a float64 rounding, and equal rationals give the same correctly rounded double -- so comparing
// cross products orders the pairs exactly as the reference's float64 scores do, without a float64 division per
// pair (it was a third of the kernel's instructions).  The double is computed once per from-string.
struct Best {
    int lcs, mx, idx;     // idx == INT_MAX: nothing yet;  both strings empty is stored as 1 / 1 (ratio 100)
};

__device__ inline bool better(int lcs_a, int mx_a, int idx_a, const Best &b)
{
    const uint64_t l = (uint64_t)(uint32_t)lcs_a * (uint32_t)b.mx, r = (uint64_t)(uint32_t)b.lcs * (uint32_t)mx_a;
    return b.idx == INT_MAX ? idx_a != INT_MAX : (l > r || (l == r && idx_a < b.idx));
}

__device__ inline void take(Best &b, int lcs, int mx, int idx)
{
    if (better(lcs, mx, idx, b)) {
        b.lcs = lcs;
        b.mx = mx;
        b.idx = idx;
    }
}

__device__ inline void wave_best(Best &b)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int ol = __shfl_xor(b.lcs, d, 64), om = __shfl_xor(b.mx, d, 64), oi = __shfl_xor(b.idx, d, 64);
        if (oi != INT_MAX) take(b, ol, om, oi);
    }
}

// rapidfuzz: norm_dist = dist / maximum (0 when both empty); ratio = (1 - norm_dist) * 100
__device__ inline double ratio_of(int lcs, int64_t maximum)
{
    const int64_t dist = maximum - 2 * (int64_t)lcs;
    const double norm_dist = maximum != 0 ? (double)dist / (double)maximum : 0.0;
    return (1.0 - norm_dist) * 100.0;
}

template <typename WORD, int W, int IDB>
__global__ __launch_bounds__(256) void k4_indel_kernel(IndelArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    WORD *pm = (WORD *)smem_raw;
    __shared__ int red[4][3];
    constexpr int WB = sizeof(WORD) * 8;
    constexpr int PER = 32 / IDB;  // symbols per dword
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int p = tid; p < A.n_sym1 * W; p += 256) pm[p] = 0;
    __syncthreads();

    const int parts = A.parts;
    for (int u = blockIdx.x; u < A.n_rows * parts; u += gridDim.x) {
        const int r = u / parts, part = u - r * parts;
        const int row = A.rows[r];
        const int64_t a0 = A.a_off[row];
        const int m = (int)(A.a_off[row + 1] - a0);
        auto a_sym = [&](int p) -> int {
            const uint32_t c = A.a_width == 1 ? (uint32_t)((const uint8_t *)A.a_chars)[a0 + p] : ((const uint32_t *)A.a_chars)[a0 + p];
            return c < A.lut_len ? (int)A.lut[c] : 0;
        };
        for (int p = tid; p < m; p += 256) {
            const int sy = a_sym(p);
            if (sy) lds_or(&pm[sy * W + p / WB], (WORD)1 << (p % WB));
        }
        __syncthreads();

        const int skip = A.skip_idx ? A.skip_idx[row] : -1;
        Best best = {0, 1, INT_MAX};
        for (int g = wave + 4 * part; g < A.n_groups; g += 4 * parts) {
            WORD V[W];
#pragma unroll
            for (int w = 0; w < W; ++w) V[w] = ~(WORD)0;
            const uint32_t *gp = A.b_packed + A.g_off[g] + lane;
            const int steps = A.g_steps[g];
#pragma unroll 2
            for (int t = 0; t < steps; ++t) {
                const uint32_t pk = gp[(int64_t)t * 64];
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const uint32_t c = (pk >> (q * IDB)) & ((1u << IDB) - 1u);
                    lcs_step<WORD, W>(V, pm + c * W);
                }
            }
            int lcs = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) lcs += popc_word((WORD)~V[w]);
            const int slot = g * 64 + lane;
            const int orig = A.b_orig[slot];
            if (orig >= 0) {
                const int lb = A.b_len[slot];
                if (A.matrix) A.matrix[((int64_t)row - A.from_begin) * A.n_to + orig] = choice_left_out(orig, skip, A.skip_up_to) ? -1.0 : ratio_of(lcs, (int64_t)m + lb);
                if (!choice_left_out(orig, skip, A.skip_up_to)) {
                    if (m + lb == 0) take(best, 1, 1, orig);
                    else take(best, lcs, m + lb, orig);
                }
            }
        }
        // first maximum: (score desc, original index asc)
        wave_best(best);
        if (lane == 0) {
            red[wave][0] = best.lcs;
            red[wave][1] = best.mx;
            red[wave][2] = best.idx;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (red[w][2] != INT_MAX) take(best, red[w][0], red[w][1], red[w][2]);
            if (parts > 1) {
                int *dst = A.partial + ((int64_t)r * parts + part) * 3;
                dst[0] = best.lcs;
                dst[1] = best.mx;
                dst[2] = best.idx;
            }
            else {
                const int64_t o = (int64_t)row - A.from_begin;
                A.out_idx[o] = best.idx == INT_MAX ? -1 : best.idx;
                // (1 / 1 can only be the stored form of two empty strings -- maximum = 1 has lcs = 0: ratio 100)
                A.out_score[o] = best.idx == INT_MAX ? 0.0 : (best.lcs == 1 && best.mx == 1 ? 100.0 : ratio_of(best.lcs, best.mx));
            }
        }
        // clear the PM entries of this from-string
        for (int p = tid; p < m; p += 256) {
            const int sy = a_sym(p);
#pragma unroll
            for (int w = 0; w < W; ++w) pm[sy * W + w] = 0;       // (symbol 0 = padding: its entry is zero anyway)
        }
        __syncthreads();
    }
}

// From-strings of <= 32 characters (95 % of the IMDB titles, 80 % of the company names) go FOUR per workgroup pass:
// the match table holds the four 32-bit masks of a symbol side by side, so one ds_read_b128 -- and one extraction of the
// to-symbol, one address, one load of the packed to-characters, one trip through the group loop -- serves four Indel
// recurrences.  From-strings of <= 16 characters (half of the IMDB titles) go EIGHT per pass (NS = 8): two 16-bit masks
// per table word and the recurrence's addition as v_pk_add_u16 (no carry between the halves; and / xor / or do not
// care) -- the same 18 vector operations per to-character then serve eight strings.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// low 32 bits of the product of the operands' low 24 bits, at full rate (v_mul_lo_u32 is quarter rate)
__device__ inline uint32_t mul_u24(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

template <int IDB, int NS>
__global__ __launch_bounds__(256, NS == 8 ? 6 : 7) void k4_indel_quad_kernel(IndelArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *pm = (uint4 *)smem_raw;
    __shared__ int red[4][NS][3];
    __shared__ int s_m[NS];
    __shared__ int64_t s_a0[NS];
    constexpr int PER = 32 / IDB;
    constexpr int CH = 128 / NS;          // characters per from-string: 32 (one mask per word) or 16 (two)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int p = tid; p < A.n_sym1; p += 256) pm[p] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    const int n_units = (A.n_rows + NS - 1) / NS, parts = A.parts;
    for (int u = blockIdx.x; u < n_units * parts; u += gridDim.x) {
        const int qd = u / parts, part = u - qd * parts;
        int row[NS], m[NS], skip[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int r = qd * NS + k;
            row[k] = r < A.n_rows ? A.rows[r] : -1;
            const int64_t a0 = row[k] >= 0 ? A.a_off[row[k]] : 0;
            m[k] = row[k] >= 0 ? (int)(A.a_off[row[k] + 1] - a0) : 0;
            skip[k] = (A.skip_idx && row[k] >= 0) ? A.skip_idx[row[k]] : -1;
            if (tid == 0) {
                s_m[k] = m[k];
                s_a0[k] = a0;
            }
        }
        __syncthreads();
        // thread t < 128 owns character t % CH of string t / CH (mask bit t % CH of word (t / CH) & 3, upper half for
        // strings 4..7)
        const int myk = (tid / CH) & (NS - 1), myp = tid & (CH - 1);
        const uint32_t my_bit = 1u << (myp + (NS == 8 ? 16 * (myk >> 2) : 0));
        int my_sym = 0;
        if (tid < 128 && myp < s_m[myk]) {
            const int64_t at = s_a0[myk] + myp;
            const uint32_t c = A.a_width == 1 ? (uint32_t)((const uint8_t *)A.a_chars)[at] : ((const uint32_t *)A.a_chars)[at];
            my_sym = c < A.lut_len ? (int)A.lut[c] : 0;
        }
        if (my_sym) atomicOr((uint32_t *)&pm[my_sym] + (myk & 3), my_bit);
        __syncthreads();

        Best best[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) best[k] = Best{0, 1, INT_MAX};
        for (int g = wave + 4 * part; g < A.n_groups; g += 4 * parts) {
            uint32_t V0 = ~0u, V1 = ~0u, V2 = ~0u, V3 = ~0u;
            // (g is wave-uniform: say so, or the loop below is compiled with a per-lane trip count and exec masks)
            const int steps = __builtin_amdgcn_readfirstlane(A.g_steps[g]);
            const int64_t g_off_v = A.g_off[g];
            const int64_t g_off = ((int64_t)__builtin_amdgcn_readfirstlane((int)(g_off_v >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)g_off_v);
            const uint32_t *gp = A.b_packed + g_off + lane;
            // V' = (V + (V & PM)) | (V & ~PM) as u = V & PM, V' = (V + u) | (V ^ u): four two-operand operations.
            // (A complemented table and t = V & ~PM, V' = ((V ^ t) + V) | t would be three, with the xor and the add
            // fused in v_xad_u32 -- but that one reads three registers and issues at half rate: 3.7 cycles per
            // instruction against 2.6, tools/ubench/valu_rate.hip; no gain, tried.)
            auto rec = [](uint32_t &V, uint32_t mask) {
                const uint32_t u = V & mask;
                // three operations: the or of the sum with V & ~PM is ONE v_bitop3_b32 (or_andnot above); 20k x 20k titles 1.036 ->
                // 0.972 ms against the four two-operand operations the compiler makes of (V + u) | (V ^ u)
                uint32_t s;
                if (NS == 8) s = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, V) + __builtin_bit_cast(u16x2, u)));
                else s = V + u;
                V = or_andnot(s, V, u);
            };
            // The packed to-characters are fetched two steps ahead of their use; the to-string's length and original
            // index (needed after the character loop) are requested before it.
            const int slot = g * 64 + lane;
            const int orig = A.b_orig[slot];
            const int lb = A.b_len[slot];
            // (reads up to two steps past the group: the next group's characters or the buffer's padding, never used)
            uint32_t pk = gp[0], pk1 = gp[64];
#pragma unroll 2
            for (int t = 0; t < steps; ++t) {
#if PFZ_K4_EXP == 3
                const uint32_t pk2 = pk * 1664525u + 1013904223u;
#else
                const uint32_t pk2 = gp[(int64_t)(t + 2) * 64];
#endif
#pragma unroll
                for (int q = 0; q < PER; ++q) {
#if PFZ_K4_EXP == 1       // (what-if builds, results wrong: 1 = no table look-up, 2 = no per-pair epilogue, 3 = no packed-character loads)
                    const uint4 N = make_uint4(pk, pk >> 3, pk * 5u, pk ^ (uint32_t)q);
#else
                    const uint4 N = pm[__builtin_amdgcn_ubfe(pk, q * IDB, IDB)];
#endif
                    rec(V0, N.x);
                    rec(V1, N.y);
                    rec(V2, N.z);
                    rec(V3, N.w);
                }
                pk = pk1;
                pk1 = pk2;
            }
#if PFZ_K4_EXP == 2
            if (orig >= 0 && (V0 ^ V1 ^ V2 ^ V3) == 0x12345u) best[0].idx = orig;
            if (false) {
#else
            if (orig >= 0) {
#endif
                const uint32_t nv[4] = {~V0, ~V1, ~V2, ~V3};
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int lcs = NS == 8 ? (k < 4 ? (int)__popc(nv[k & 3] & 0xffffu) : (int)__popc(nv[k & 3] >> 16)) : (int)__popc(nv[k & 3]);
                    // lcs <= 32 and |a| + |b| < 2^24 here (checked at the launch): the cross products of better() fit
                    // 32 bits and v_mul_u32_u24 forms them at full rate (two 64-bit products per pair were a third of
                    // this kernel's time)
                    // (the initial best 0 / 1 with index INT_MAX loses to every candidate -- equal products, lower
                    // index -- so "nothing yet" needs no test; slots beyond the last row are never written out)
                    int l_a = lcs, m_a = m[k] + lb;
                    // (an empty from-string: "" vs "" is 1 / 1 = ratio 100.  Wave-uniform, and said so: as a select it was two
                    // vector instructions per pair in a kernel that is bound by their rate)
                    if (__builtin_amdgcn_readfirstlane(m[k]) == 0) {
                        l_a = lb == 0 ? 1 : 0;
                        m_a = lb == 0 ? 1 : lb;
                    }
                    // branch-free on purpose: with && / || the compiler builds a maze of exec-mask branches here
                    // (the per-pair epilogue was 28 % of the kernel, what-if build 2)
                    const uint32_t l = mul_u24((uint32_t)l_a, (uint32_t)best[k].mx), r = mul_u24((uint32_t)best[k].lcs, (uint32_t)m_a);
                    bool wins = (l > r) | ((l == r) & (orig < best[k].idx));
                    if (A.skip_idx) wins = wins & (A.skip_up_to ? orig > skip[k] : orig != skip[k]);
                    best[k].lcs = wins ? l_a : best[k].lcs;
                    best[k].mx = wins ? m_a : best[k].mx;
                    best[k].idx = wins ? orig : best[k].idx;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            wave_best(best[k]);
            if (lane == 0) {
                red[wave][k][0] = best[k].lcs;
                red[wave][k][1] = best[k].mx;
                red[wave][k][2] = best[k].idx;
            }
        }
        __syncthreads();
        if (tid < NS) {
            const int k = tid, r = qd * NS + k;
            if (r < A.n_rows) {
                Best b = {red[0][k][0], red[0][k][1], red[0][k][2]};
                for (int w = 1; w < 4; ++w)
                    if (red[w][k][2] != INT_MAX) take(b, red[w][k][0], red[w][k][1], red[w][k][2]);
                if (parts > 1) {
                    int *dst = A.partial + ((int64_t)r * parts + part) * 3;
                    dst[0] = b.lcs;
                    dst[1] = b.mx;
                    dst[2] = b.idx;
                }
                else {
                    const int64_t o = (int64_t)A.rows[r] - A.from_begin;
                    A.out_idx[o] = b.idx == INT_MAX ? -1 : b.idx;
                    A.out_score[o] = b.idx == INT_MAX ? 0.0 : (b.lcs == 1 && b.mx == 1 ? 100.0 : ratio_of(b.lcs, b.mx));
                }
            }
        }
        if (my_sym) atomicAnd((uint32_t *)&pm[my_sym] + (myk & 3), ~my_bit);      // clear the entries this unit set
        __syncthreads();
    }
}

// parts > 1: the best of a from-string over its parts (same order: score desc, original index asc) and its score
__global__ __launch_bounds__(256) void k4_merge_parts(IndelArgs A)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= A.n_rows) return;
    Best b = {0, 1, INT_MAX};
    for (int p = 0; p < A.parts; ++p) {
        const int *src = A.partial + ((int64_t)r * A.parts + p) * 3;
        if (src[2] != INT_MAX) take(b, src[0], src[1], src[2]);
    }
    const int64_t o = (int64_t)A.rows[r] - A.from_begin;
    A.out_idx[o] = b.idx == INT_MAX ? -1 : b.idx;
    A.out_score[o] = b.idx == INT_MAX ? 0.0 : (b.lcs == 1 && b.mx == 1 ? 100.0 : ratio_of(b.lcs, b.mx));
}

// The general case: from-strings beyond 1024 characters, or an alphabet x word count whose match table does not fit
// LDS (thousands of distinct CJK characters with long strings).  Same recurrence, any number W of 64-bit words, with
// the match table of the workgroup's from-string and every lane's V in global scratch (L2): slow -- every word-step
// is two loads and a store -- but the reference accepts such inputs, so the engine does too.  One to-string per lane.
template <int IDB>
__global__ __launch_bounds__(256) void k4_indel_general_kernel(IndelArgs A, int32_t W, uint64_t *__restrict__ pm_all,
                                                                uint64_t *__restrict__ v_all)
{
    __shared__ int red[4][3];
    constexpr int PER = 32 / IDB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t *pm = pm_all + (int64_t)blockIdx.x * A.n_sym1 * W;      // zero on entry, zero again after every row
    uint64_t *V = v_all + (int64_t)blockIdx.x * W * 256 + tid;        // V[w * 256]: this lane's word w
    for (int r = blockIdx.x; r < A.n_rows; r += gridDim.x) {
        const int row = A.rows[r];
        const int64_t a0 = A.a_off[row];
        const int m = (int)(A.a_off[row + 1] - a0);
        auto a_sym = [&](int p) -> int {
            const uint32_t c = A.a_width == 1 ? (uint32_t)((const uint8_t *)A.a_chars)[a0 + p] : ((const uint32_t *)A.a_chars)[a0 + p];
            return c < A.lut_len ? (int)A.lut[c] : 0;
        };
        for (int p = tid; p < m; p += 256) {
            const int sy = a_sym(p);
            if (sy) atomicOr((unsigned long long *)&pm[(int64_t)sy * W + p / 64], 1ull << (p % 64));
        }
        __threadfence_block();
        __syncthreads();
        const int skip = A.skip_idx ? A.skip_idx[row] : -1;
        Best best = {0, 1, INT_MAX};
        for (int g = wave; g < A.n_groups; g += 4) {
            for (int w = 0; w < W; ++w) V[(int64_t)w * 256] = ~0ull;
            const uint32_t *gp = A.b_packed + A.g_off[g] + lane;
            const int steps = A.g_steps[g];
            for (int t = 0; t < steps; ++t) {
                const uint32_t pk = gp[(int64_t)t * 64];
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const uint32_t c = (pk >> (q * IDB)) & ((1u << IDB) - 1u);
                    if (c == 0) continue;                           // padding symbol: an empty mask changes nothing
                    const uint64_t *pmc = pm + (int64_t)c * W;
                    uint64_t carry = 0;
                    for (int w = 0; w < W; ++w) {
                        const uint64_t M = pmc[w], sv = V[(int64_t)w * 256], u = sv & M;
                        uint64_t sum = sv + u;
                        const uint64_t c1 = sum < sv ? 1 : 0;
                        sum += carry;
                        const uint64_t c2 = sum < carry ? 1 : 0;
                        carry = c1 | c2;
                        V[(int64_t)w * 256] = sum | (sv ^ u);
                    }
                }
            }
            int lcs = 0;
            for (int w = 0; w < W; ++w) lcs += __popcll(~V[(int64_t)w * 256]);
            const int slot = g * 64 + lane;
            const int orig = A.b_orig[slot];
            if (orig >= 0) {
                const int lb = A.b_len[slot];
                if (A.matrix) A.matrix[((int64_t)row - A.from_begin) * A.n_to + orig] = choice_left_out(orig, skip, A.skip_up_to) ? -1.0 : ratio_of(lcs, (int64_t)m + lb);
                if (!choice_left_out(orig, skip, A.skip_up_to)) {
                    if (m + lb == 0) take(best, 1, 1, orig);
                    else take(best, lcs, m + lb, orig);
                }
            }
        }
        wave_best(best);
        if (lane == 0) {
            red[wave][0] = best.lcs;
            red[wave][1] = best.mx;
            red[wave][2] = best.idx;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (red[w][2] != INT_MAX) take(best, red[w][0], red[w][1], red[w][2]);
            const int64_t o = (int64_t)row - A.from_begin;
            A.out_idx[o] = best.idx == INT_MAX ? -1 : best.idx;
            A.out_score[o] = best.idx == INT_MAX ? 0.0 : (best.lcs == 1 && best.mx == 1 ? 100.0 : ratio_of(best.lcs, best.mx));
        }
        for (int p = tid; p < m; p += 256) {
            const int sy = a_sym(p);
            if (sy) pm[(int64_t)sy * W + p / 64] = 0ull;
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ---- to-side plan (cached on the to-list's handle) -----------------------------

// distinct code units of a list: LDS bitmap for code points < 65536, global atomics beyond
template <int CW>
__global__ __launch_bounds__(256) void k4_mark_alphabet(const void *__restrict__ chars, int64_t n_units, uint32_t *__restrict__ present)
{
    __shared__ uint32_t bm[2048];
    for (int t = threadIdx.x; t < 2048; t += 256) bm[t] = 0u;
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_units; p += (int64_t)gridDim.x * 256) {
        const uint32_t c = CW == 1 ? (uint32_t)((const uint8_t *)chars)[p] : ((const uint32_t *)chars)[p];
        if (c < 65536u) {
            if (!((bm[c >> 5] >> (c & 31)) & 1u)) atomicOr(&bm[c >> 5], 1u << (c & 31));
        } else if (c < 0x110000u) {
            atomicOr(&present[c >> 5], 1u << (c & 31));
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2048; t += 256)
        if (bm[t]) atomicOr(&present[t], bm[t]);
}

// slot (group g, lane l) = to-string order[g*64 + l]: its symbols as [t/PER][lane] dwords
template <int CW, int IDB>
__global__ __launch_bounds__(256) void k4_pack(const void *__restrict__ chars, const int64_t *__restrict__ off,
                                                const int32_t *__restrict__ b_orig, const int64_t *__restrict__ g_off,
                                                const int32_t *__restrict__ g_steps, int64_t n_slots,
                                                const uint16_t *__restrict__ lut, uint32_t lut_len, uint32_t *__restrict__ packed)
{
    constexpr int PER = 32 / IDB;
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= n_slots) return;
    const int64_t g = slot >> 6;
    const int lane = (int)(slot & 63);
    const int steps = g_steps[g];
    uint32_t *dst = packed + g_off[g] + lane;
    const int32_t j = b_orig[slot];
    const int64_t b0 = j >= 0 ? off[j] : 0;
    const int len = j >= 0 ? (int)(off[j + 1] - b0) : 0;
    for (int t = 0; t < steps; ++t) {
        uint32_t pk = 0u;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int pos = t * PER + q;
            if (pos < len) {
                const uint32_t c = CW == 1 ? (uint32_t)((const uint8_t *)chars)[b0 + pos] : ((const uint32_t *)chars)[b0 + pos];
                pk |= (uint32_t)(c < lut_len ? lut[c] : 0) << (q * IDB);
            }
        }
        dst[(int64_t)t * 64] = pk;
    }
}

}  // namespace pfz

struct pfz_indel_plan {
    pfz_ctx *ctx = nullptr;
    int32_t n_sym = 0, idb = 8;          // alphabet size, bits per packed symbol
    uint32_t lut_len = 0;
    uint16_t *lut = nullptr;             // device [lut_len]
    uint32_t *packed = nullptr;          // device
    int64_t *g_off = nullptr;
    int32_t *g_steps = nullptr, *b_len = nullptr, *b_orig = nullptr;
    int64_t n_groups = 0;
    int64_t char_steps = 0;              // sum over to-strings of their (padded) steps * 64 / per: the bench's work count
    ~pfz_indel_plan()
    {
        for (void *p : {(void *)lut, (void *)packed, (void *)g_off, (void *)g_steps, (void *)b_len, (void *)b_orig})
            if (p) pfz::pool_free(p);
    }
};

void pfz_indel_plan_free(pfz_indel_plan *p) { delete p; }

namespace pfz {

template <typename T> static int up(pfz_ctx *ctx, T **dst, const std::vector<T> &v)
{
    PFZ_TRY(pool_alloc(ctx, dst, (v.empty() ? 1 : v.size()) * sizeof(T)));
    if (!v.empty()) PFZ_TRY(copy_h2d(ctx, *dst, v.data(), v.size() * sizeof(T)));
    return PFZ_OK;
}

static int build_plan(pfz_ctx *ctx, const pfz_strings *T, pfz_indel_plan **out)
{
    Owner<pfz_indel_plan, pfz_indel_plan_free> pl(new pfz_indel_plan());
    pl->ctx = ctx;
    // alphabet of the to-list: presence bitmap on the device, ranks on the host
    const size_t words = 0x110000 / 32;
    uint32_t *present = nullptr;
    PFZ_TRY(pool_alloc(ctx, &present, words * sizeof(uint32_t)));
    struct Free {
        void *p;
        ~Free() { pool_free(p); }
    } free_present{present};
    const size_t used_words = T->char_width == 1 ? 8 : words;
    PFZ_HIP(hipMemsetAsync(present, 0, used_words * sizeof(uint32_t), ctx->stream));
    if (T->n_units > 0) {
        const unsigned grid = (unsigned)std::min<int64_t>((T->n_units + 255) / 256, 2048);
        if (T->char_width == 1)
            hipLaunchKernelGGL(k4_mark_alphabet<1>, dim3(grid), dim3(256), 0, ctx->stream, T->chars, T->n_units, present);
        else
            hipLaunchKernelGGL(k4_mark_alphabet<4>, dim3(grid), dim3(256), 0, ctx->stream, T->chars, T->n_units, present);
    }
    std::vector<uint32_t> h(used_words);
    PFZ_TRY(copy_d2h(ctx, h.data(), present, used_words * sizeof(uint32_t)));
    std::vector<uint16_t> lut;
    int32_t S = 0;
    for (size_t wi = used_words; wi-- > 0;)
        if (h[wi]) {
            lut.assign((wi + 1) * 32, 0);
            break;
        }
    for (size_t wi = 0; wi * 32 < lut.size(); ++wi) {
        uint32_t word = h[wi];
        while (word) {
            const int bit = __builtin_ctz(word);
            word &= word - 1;
            if (S >= 65535) {
                set_error("pfz_indel: more than 65535 distinct code points in the to-list exceed the 16-bit symbol space");
                return PFZ_ERR_UNSUPPORTED;
            }
            lut[wi * 32 + (size_t)bit] = (uint16_t)(++S);
        }
    }
    pl->n_sym = S;
    pl->idb = S <= 255 ? 8 : 16;
    pl->lut_len = (uint32_t)lut.size();
    PFZ_TRY(up(ctx, &pl->lut, lut));
    // groups of 64 to-strings of similar length: counting sort by length on the host (O(n) ints), packing on the device
    const int per = 32 / pl->idb;
    const int64_t n_to = T->n;
    std::vector<int64_t> start((size_t)T->max_len + 2, 0);
    for (int64_t j = 0; j < n_to; ++j) start[(size_t)(T->h_off[(size_t)j + 1] - T->h_off[(size_t)j]) + 1]++;
    for (size_t l = 1; l < start.size(); ++l) start[l] += start[l - 1];
    const int64_t n_groups = (n_to + 63) / 64;
    std::vector<int32_t> b_len((size_t)n_groups * 64, 0), b_orig((size_t)n_groups * 64, -1), g_steps((size_t)n_groups);
    std::vector<int64_t> g_off((size_t)n_groups);
    for (int64_t j = 0; j < n_to; ++j) {      // ascending j inside one length: a stable sort
        const int64_t len = T->h_off[(size_t)j + 1] - T->h_off[(size_t)j];
        const int64_t pos = start[(size_t)len]++;
        b_len[(size_t)pos] = (int32_t)len;
        b_orig[(size_t)pos] = (int32_t)j;
    }
    int64_t total_dw = 0;
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t last = std::min<int64_t>(n_to, (g + 1) * 64) - 1;     // sorted: the group's longest string
        g_off[(size_t)g] = total_dw;
        g_steps[(size_t)g] = (int32_t)((b_len[(size_t)last] + per - 1) / per);
        total_dw += (int64_t)g_steps[(size_t)g] * 64;
    }
    pl->n_groups = n_groups;
    pl->char_steps = total_dw * per;
    PFZ_TRY(up(ctx, &pl->g_off, g_off));
    PFZ_TRY(up(ctx, &pl->g_steps, g_steps));
    PFZ_TRY(up(ctx, &pl->b_len, b_len));
    PFZ_TRY(up(ctx, &pl->b_orig, b_orig));
    PFZ_TRY(pool_alloc(ctx, &pl->packed, (size_t)(total_dw + 128) * sizeof(uint32_t)));     // + two steps: the quad kernel reads ahead
    if (n_groups > 0) {
        ProfScope ps(ctx, "k4_pack");
        const unsigned grid = (unsigned)((n_groups * 64 + 255) / 256);
#define PFZ_K4_PACK(CW, IDB)                                                                                        \
    hipLaunchKernelGGL((k4_pack<CW, IDB>), dim3(grid), dim3(256), 0, ctx->stream, T->chars, T->offsets, pl->b_orig, \
                       pl->g_off, pl->g_steps, n_groups * 64, pl->lut, pl->lut_len, pl->packed)
        if (T->char_width == 1 && pl->idb == 8) PFZ_K4_PACK(1, 8);
        else if (T->char_width == 1) PFZ_K4_PACK(1, 16);
        else if (pl->idb == 8) PFZ_K4_PACK(4, 8);
        else PFZ_K4_PACK(4, 16);
#undef PFZ_K4_PACK
        PFZ_HIP(hipGetLastError());
    }
    *out = pl.release();
    return PFZ_OK;
}

template <typename WORD, int W>
static int launch_class(pfz_ctx *ctx, const IndelArgs &A, int idb, unsigned grid, hipStream_t st)
{
    const size_t lds = (size_t)A.n_sym1 * W * sizeof(WORD);
    if (idb == 8)
        hipLaunchKernelGGL((k4_indel_kernel<WORD, W, 8>), dim3(grid), dim3(256), lds, st, A);
    else
        hipLaunchKernelGGL((k4_indel_kernel<WORD, W, 16>), dim3(grid), dim3(256), lds, st, A);
    PFZ_HIP(hipGetLastError());
    return PFZ_OK;
}

struct DevBuf {
    pfz_ctx *ctx = nullptr;
    void *p = nullptr;
    ~DevBuf() { if (p) pool_free(p); }
    int alloc(size_t bytes) { return pool_alloc(ctx, &p, bytes > 0 ? bytes : 16); }
};

static int indel_run(pfz_ctx *ctx, const pfz_strings *F, const pfz_strings *T_c, const int32_t *skip_idx, int64_t begin,
                     int64_t end, int32_t *out_idx, double *out_score, double *out_matrix, pfz_topn *out_dev = nullptr)
{
    PFZ_REQUIRE(ctx && F && T_c, "pfz_indel: NULL argument");
    PFZ_REQUIRE(begin >= 0 && begin <= end && end <= F->n, "pfz_indel: row range [%lld,%lld) outside [0,%lld)",
                (long long)begin, (long long)end, (long long)F->n);
    const int64_t n_rows = end - begin;
    if (n_rows == 0) return PFZ_OK;
    PFZ_HIP(hipSetDevice(ctx->device));
    if (T_c->n >= INT_MAX - 64 || F->n >= INT_MAX) {
        set_error("pfz_indel: list too long");
        return PFZ_ERR_UNSUPPORTED;
    }
    pfz_strings *T = const_cast<pfz_strings *>(T_c);    // the plan cache lives inside the (otherwise read-only) to-list
    if (!T->indel_plan) PFZ_TRY(build_plan(ctx, T, &T->indel_plan));
    const pfz_indel_plan *pl = T->indel_plan;

    // from side: word classes by length
    static const int kClassMax[7] = {32, 64, 128, 256, 512, 1024, INT_MAX};
    std::vector<int32_t> cls[7];
    for (int64_t i = begin; i < end; ++i) {
        const int64_t m = F->h_off[(size_t)i + 1] - F->h_off[(size_t)i];
        int c = 0;
        while (m > kClassMax[c]) ++c;
        cls[c].push_back((int32_t)i);
    }
    // classes whose match table (alphabet x words) does not fit 60 KiB of LDS join the general class 6 (> 1024 characters)
    static const int kClassWordBytes[6] = {4, 8, 16, 32, 64, 128};
    for (int c = 0; c < 6; ++c)
        if ((size_t)(pl->n_sym + 1) * kClassWordBytes[c] > 60 * 1024 && !cls[c].empty()) {
            cls[6].insert(cls[6].end(), cls[c].begin(), cls[c].end());
            cls[c].clear();
        }
    if (getenv("PFZ_K4_FORCE_GENERAL")) {       // tests: everything through the general kernel
        for (int c = 0; c < 6; ++c) {
            cls[6].insert(cls[6].end(), cls[c].begin(), cls[c].end());
            cls[c].clear();
        }
    }
    const int64_t n_to = T->n;
    DevBuf d_skip, d_oidx, d_oscore, d_matrix, d_rows[7], d_pm, d_v;
    for (DevBuf *b : {&d_skip, &d_oidx, &d_oscore, &d_matrix, &d_rows[0], &d_rows[1], &d_rows[2], &d_rows[3], &d_rows[4], &d_rows[5],
                      &d_rows[6], &d_pm, &d_v})
        b->ctx = ctx;
    int skip_up_to = 0;
    if (skip_idx) {
        std::vector<int32_t> codes(skip_idx, skip_idx + F->n);
        skip_up_to = decode_skip_codes(codes);
        PFZ_REQUIRE(skip_up_to >= 0, "pfz_indel_argmax: skip_idx mixes single choices (>= 0) and 'up to' codes (<= -2)");
        PFZ_TRY(d_skip.alloc((size_t)F->n * sizeof(int32_t)));
        PFZ_TRY(copy_h2d(ctx, d_skip.p, codes.data(), (size_t)F->n * sizeof(int32_t)));
    }
    PFZ_TRY(d_oidx.alloc((size_t)n_rows * sizeof(int32_t)));
    PFZ_TRY(d_oscore.alloc((size_t)n_rows * sizeof(double)));
    if (out_matrix) PFZ_TRY(d_matrix.alloc((size_t)n_rows * (size_t)n_to * sizeof(double)));

    IndelArgs A;
    A.a_chars = F->chars;
    A.a_width = F->char_width;
    A.a_off = F->offsets;
    A.lut = pl->lut;
    A.lut_len = pl->lut_len;
    A.b_packed = pl->packed;
    A.g_off = pl->g_off;
    A.g_steps = pl->g_steps;
    A.b_len = pl->b_len;
    A.b_orig = pl->b_orig;
    A.n_groups = (int32_t)pl->n_groups;
    A.skip_idx = skip_idx ? (const int32_t *)d_skip.p : nullptr;
    A.skip_up_to = skip_up_to;
    A.n_sym1 = pl->n_sym + 1;
    A.from_begin = begin;
    A.n_to = n_to;
    A.out_idx = (int32_t *)d_oidx.p;
    A.out_score = (double *)d_oscore.p;
    A.matrix = out_matrix ? (double *)d_matrix.p : nullptr;
    const int64_t max_grid = (int64_t)ctx->prop.multiProcessorCount * 8;
    A.parts = 1;
    A.partial = nullptr;
    // A class with few rows (the 33 IMDB titles beyond 64 characters ran as 33 workgroups for 178 us) or a unit count
    // that is a small non-integer multiple of the chip's workgroup slots (5000 quads on 2048 slots: the third round is
    // 44 % full) splits every unit's to-groups over `parts` workgroups: >= 4 rounds of work units, each part at least
    // one group per wave.  The parts' bests meet in k4_merge_parts.
    auto split = [&](int64_t n_units) {
        const int64_t want = (4 * max_grid + n_units - 1) / n_units, cap = std::max<int64_t>(1, pl->n_groups / 4);
        int64_t parts = std::max<int64_t>(1, std::min(want, cap));
        if (const char *e = getenv("PFZ_K4_PARTS")) parts = std::max(1, atoi(e));      // tests, A/B timing
        return (int32_t)parts;
    };
    auto merge = [&](hipStream_t st) -> int {
        if (A.parts <= 1) return PFZ_OK;
        hipLaunchKernelGGL(k4_merge_parts, dim3((unsigned)((A.n_rows + 255) / 256)), dim3(256), 0, st, A);
        PFZ_HIP(hipGetLastError());
        return PFZ_OK;
    };
    // every launch has its own (row, part) records: the classes run on two streams
    DevBuf d_part[8];
    int n_part = 0;
    auto ensure_partial = [&]() -> int {
        A.partial = nullptr;
        if (A.parts <= 1) return PFZ_OK;
        DevBuf &b = d_part[n_part++];
        b.ctx = ctx;
        PFZ_TRY(b.alloc((size_t)A.n_rows * (size_t)A.parts * 3 * sizeof(int32_t)));
        A.partial = (int32_t *)b.p;
        return PFZ_OK;
    };
    // PFZ_K4_SIDE_STREAM=1: the classes of longer from-strings (few rows each: 5 % of the IMDB titles, a fifth of the kernel
    // time) on the side stream, beside the two big launches of class 0.  Measured on 20k x 20k titles: the kernels overlap
    // (1.02 -> 0.97 ms), but the two cross-stream dependencies cost 0.4 ms of host / queue time per call (step 1.06 -> 1.45 ms):
    // off by default.  (K7 uses the same scheme where the overlapped launch is 7 ms long.)
    bool any_long = false;
    for (int c = 1; c < 6; ++c) any_long = any_long || !cls[c].empty();
    const bool side = any_long && !cls[0].empty() && getenv("PFZ_K4_SIDE_STREAM") != nullptr;
    if (side) PFZ_TRY(ensure_side_stream(ctx));
    // the row lists of all classes first (the side stream starts from an event recorded behind these copies); class 0:
    // several short from-strings per workgroup pass (PFZ_K4_NO_QUAD=1: the one-string kernel, tests) -- eight of <= 16
    // characters, four of 17 .. 32 (PFZ_K4_NO_OCTO=1: four of <= 32)
    const bool quad = !cls[0].empty() && !out_matrix && (size_t)A.n_sym1 * sizeof(uint4) <= 60 * 1024 && T->max_len < (1 << 24) - 64 &&
                      !getenv("PFZ_K4_NO_QUAD");
    std::vector<int32_t> rows8, rows4;
    if (quad) {
        for (int32_t i : cls[0])
            (F->h_off[(size_t)i + 1] - F->h_off[(size_t)i] <= 16 && !getenv("PFZ_K4_NO_OCTO") ? rows8 : rows4).push_back(i);
        std::copy(rows4.begin(), rows4.end(), std::copy(rows8.begin(), rows8.end(), cls[0].begin()));
    }
    for (int c = 0; c < 6; ++c) {
        if (cls[c].empty()) continue;
        PFZ_TRY(d_rows[c].alloc(cls[c].size() * sizeof(int32_t)));
        PFZ_TRY(copy_h2d(ctx, d_rows[c].p, cls[c].data(), cls[c].size() * sizeof(int32_t)));
    }
    ProfScope ps_all(ctx, "k4_indel");
    if (side) {
        PFZ_HIP(hipEventRecord(ctx->side_events[0], ctx->stream));      // (the inputs are ready)
        PFZ_HIP(hipStreamWaitEvent(ctx->stream2, ctx->side_events[0], 0));
    }
    for (int c = 5; c >= 0; --c) {
        if (cls[c].empty()) continue;
        hipStream_t st = (c > 0 && side) ? ctx->stream2 : ctx->stream;
        A.rows = (const int32_t *)d_rows[c].p;
        A.n_rows = (int32_t)cls[c].size();
        if (c == 0 && quad) {
            const size_t lds = (size_t)A.n_sym1 * sizeof(uint4);
            for (int pass = 0; pass < 2; ++pass) {
                const int ns = pass == 0 ? 8 : 4;
                A.rows = (const int32_t *)d_rows[0].p + (pass == 0 ? 0 : rows8.size());
                A.n_rows = (int32_t)(pass == 0 ? rows8.size() : rows4.size());
                if (A.n_rows == 0) continue;
                const int64_t n_units = (A.n_rows + ns - 1) / ns;
                A.parts = split(n_units);
                PFZ_TRY(ensure_partial());
                const dim3 qgrid((unsigned)std::min<int64_t>(n_units * A.parts, max_grid));
                if (pl->idb == 8 && ns == 8) hipLaunchKernelGGL((k4_indel_quad_kernel<8, 8>), qgrid, dim3(256), lds, st, A);
                else if (pl->idb == 8) hipLaunchKernelGGL((k4_indel_quad_kernel<8, 4>), qgrid, dim3(256), lds, st, A);
                else if (ns == 8) hipLaunchKernelGGL((k4_indel_quad_kernel<16, 8>), qgrid, dim3(256), lds, st, A);
                else hipLaunchKernelGGL((k4_indel_quad_kernel<16, 4>), qgrid, dim3(256), lds, st, A);
                PFZ_HIP(hipGetLastError());
                PFZ_TRY(merge(st));
            }
            continue;
        }
        A.parts = split(A.n_rows);
        PFZ_TRY(ensure_partial());
        const unsigned grid = (unsigned)std::min<int64_t>((int64_t)A.n_rows * A.parts, max_grid);
        switch (c) {
        case 0: PFZ_TRY((launch_class<uint32_t, 1>(ctx, A, pl->idb, grid, st))); break;
        case 1: PFZ_TRY((launch_class<uint64_t, 1>(ctx, A, pl->idb, grid, st))); break;
        case 2: PFZ_TRY((launch_class<uint64_t, 2>(ctx, A, pl->idb, grid, st))); break;
        case 3: PFZ_TRY((launch_class<uint64_t, 4>(ctx, A, pl->idb, grid, st))); break;
        case 4: PFZ_TRY((launch_class<uint64_t, 8>(ctx, A, pl->idb, grid, st))); break;
        default: PFZ_TRY((launch_class<uint64_t, 16>(ctx, A, pl->idb, grid, st))); break;
        }
        PFZ_TRY(merge(st));
    }
    if (side) {
        PFZ_HIP(hipEventRecord(ctx->side_events[1], ctx->stream2));
        PFZ_HIP(hipStreamWaitEvent(ctx->stream, ctx->side_events[1], 0));
    }
    A.parts = 1;
    if (!cls[6].empty()) {
        // the general kernel: W words for the longest of these strings, match tables and V columns in global scratch
        int64_t longest = 1;
        for (int32_t i : cls[6]) longest = std::max<int64_t>(longest, F->h_off[(size_t)i + 1] - F->h_off[(size_t)i]);
        const int32_t W = (int32_t)((longest + 63) / 64);
        int64_t gridg = std::min<int64_t>((int64_t)cls[6].size(), ctx->prop.multiProcessorCount * 2);
        const size_t pm_per = (size_t)(pl->n_sym + 1) * (size_t)W * sizeof(uint64_t);
        while (gridg > 1 && pm_per * (size_t)gridg > ((size_t)2 << 30)) gridg /= 2;      // <= 2 GiB of match tables
        if (pm_per * (size_t)gridg > ((size_t)8 << 30)) {
            set_error("pfz_indel: a from-string of %lld characters with %d alphabet symbols needs a %zu-byte match table",
                      (long long)longest, pl->n_sym, pm_per);
            return PFZ_ERR_UNSUPPORTED;
        }
        PFZ_TRY(d_pm.alloc(pm_per * (size_t)gridg));
        PFZ_TRY(d_v.alloc((size_t)gridg * (size_t)W * 256 * sizeof(uint64_t)));
        PFZ_HIP(hipMemsetAsync(d_pm.p, 0, pm_per * (size_t)gridg, ctx->stream));
        PFZ_TRY(d_rows[6].alloc(cls[6].size() * sizeof(int32_t)));
        PFZ_TRY(copy_h2d(ctx, d_rows[6].p, cls[6].data(), cls[6].size() * sizeof(int32_t)));
        A.rows = (const int32_t *)d_rows[6].p;
        A.n_rows = (int32_t)cls[6].size();
        if (pl->idb == 8)
            hipLaunchKernelGGL((k4_indel_general_kernel<8>), dim3((unsigned)gridg), dim3(256), 0, ctx->stream, A, W,
                               (uint64_t *)d_pm.p, (uint64_t *)d_v.p);
        else
            hipLaunchKernelGGL((k4_indel_general_kernel<16>), dim3((unsigned)gridg), dim3(256), 0, ctx->stream, A, W,
                               (uint64_t *)d_pm.p, (uint64_t *)d_v.p);
        PFZ_HIP(hipGetLastError());
    }
    if (out_dev) return best_to_topn(ctx, (const int32_t *)d_oidx.p, (const double *)d_oscore.p, n_rows, out_dev);     // (no copy, no wait)
    if (out_idx) PFZ_TRY(copy_d2h(ctx, out_idx, d_oidx.p, (size_t)n_rows * sizeof(int32_t)));
    if (out_score) PFZ_TRY(copy_d2h(ctx, out_score, d_oscore.p, (size_t)n_rows * sizeof(double)));
    if (out_matrix) PFZ_TRY(copy_d2h(ctx, out_matrix, d_matrix.p, (size_t)n_rows * (size_t)n_to * sizeof(double)));
    PFZ_HIP(hipStreamSynchronize(ctx->stream));
    return PFZ_OK;
}

}  // namespace pfz

using namespace pfz;

extern "C" {

int pfz_indel_argmax(pfz_ctx *ctx, const pfz_strings *from_strings, const pfz_strings *to_strings,
                     const int32_t *skip_idx, int64_t from_begin, int64_t from_end, int32_t *out_idx, double *out_score)
{
    PFZ_REQUIRE(out_idx && out_score, "pfz_indel_argmax: NULL output");
    return indel_run(ctx, from_strings, to_strings, skip_idx, from_begin, from_end, out_idx, out_score, nullptr);
}

int pfz_indel_argmax_dev(pfz_ctx *ctx, const pfz_strings *from_strings, const pfz_strings *to_strings,
                         const int32_t *skip_idx, int64_t from_begin, int64_t from_end, pfz_topn *out)
{
    PFZ_REQUIRE(out && out->ntop == 2 && out->n_rows >= from_end - from_begin,
                "pfz_indel_argmax_dev: the result buffer must have 2 columns and >= %lld rows", (long long)(from_end - from_begin));
    return indel_run(ctx, from_strings, to_strings, skip_idx, from_begin, from_end, nullptr, nullptr, nullptr, out);
}

int pfz_indel_matrix_host(pfz_ctx *ctx, const pfz_strings *from_strings, const pfz_strings *to_strings,
                          int64_t from_begin, int64_t from_end, double *out_matrix)
{
    PFZ_REQUIRE(out_matrix, "pfz_indel_matrix_host: NULL output");
    if (to_strings && to_strings->n == 0) return PFZ_OK;
    return indel_run(ctx, from_strings, to_strings, nullptr, from_begin, from_end, nullptr, nullptr, out_matrix);
}

int pfz_indel_plan_info(pfz_ctx *ctx, const pfz_strings *to_strings, int64_t *n_symbols, int64_t *n_groups, int64_t *char_steps)
{
    PFZ_REQUIRE(ctx && to_strings, "pfz_indel_plan_info: NULL argument");
    pfz_strings *T = const_cast<pfz_strings *>(to_strings);
    if (!T->indel_plan) {
        PFZ_HIP(hipSetDevice(ctx->device));
        PFZ_TRY(build_plan(ctx, T, &T->indel_plan));
    }
    if (n_symbols) *n_symbols = T->indel_plan->n_sym;
    if (n_groups) *n_groups = T->indel_plan->n_groups;
    if (char_steps) *char_steps = T->indel_plan->char_steps;
    return PFZ_OK;
}

}  // extern "C"
