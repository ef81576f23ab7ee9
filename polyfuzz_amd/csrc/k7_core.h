// K7's per-pair arithmetic: the rapidfuzz.fuzz scorers that build a different string pair for every (from, to) --
// partial_ratio, token_set_ratio, token_ratio, partial_token_*_ratio, WRatio (reference polyfuzz/models/_rapidfuzz.py:45-58,
// 106-108) -- as the exact score of ONE pair and as a cheap UPPER BOUND of it.
//
// Everything here is a plain function of one from-string's tables and one to-string's arrays: no wave-level operation,
// no shared memory, no global state -- k7_fuzz.hip calls it from its lanes, and tests/k7_core_host.cpp compiles the same
// header with g++ to hold both functions against the oracle on the CPU build box (test infrastructure; the product has
// no host path).
//
// Exact score.  Every scorer is the Indel similarity of a sub-string / sub-token-set of three per-string forms
//   form 0  the string;   form 1  its whitespace tokens sorted and joined by one space (token_sort);
//   form 2  its DISTINCT tokens sorted and joined (token_set), with token ids (equal ids <=> equal tokens), lengths
// computed with the bit-parallel LCS recurrence  V' = (V + (V & PM)) | (V & ~PM)  over 64-bit words (W per form):
//   * a SUB-RANGE of the from-form is matched by masking PM, and the LCS against every PREFIX of it is the number of
//     zero bits of V below the prefix length -- one pass over the to-string scores all windows of the from-string
//     that start at one position (partial_ratio, from-string the longer one);
//   * windows of the TO-string are separate passes, the prefixes falling out of the first pass step by step;
//   * token_set's differences: a mask over the from-form's token positions and a per-character skip on the to-side
//     (a tag per character: token number, separating space or not).
// Scores are float64 with rapidfuzz's two normalisations kept apart ((1 - dist / lensum) * 100 for ratio-like values,
// 100 - 100 dist / lensum inside token_set_ratio); windows compare as exact rationals.
// `cur` (a score some valid choice of this from-string already has) lets a COMPONENT of the score be left out when an
// upper bound of it is below `cur`: process.extractOne keeps only the maximum, so a pair that can win wins through
// another component; every bound goes through the same monotone float64 expressions as the value it bounds.
//
// Upper bound (k7_upper_bound).  U = sum over character classes c of min(count_a(c), count_b(c)) >= LCS of ANY two
// sub-multisets of the strings' characters, hence of every form, window and token difference.  With per-class counts
// packed four to a dword, U = (|a| + |b| - SAD(hist_a, hist_b)) / 2 -- one v_sad_u8 per four classes.  The bound is
// float32 and the caller prunes a pair only when bound + slack < cur, so float32 rounding never decides.
//
// PARITY UNPINNED (rapidfuzz is not installable): the oracle is oracle/fuzz_scorers.{py,c}.
#pragma once

#include <stdint.h>

#ifndef PFZ_HD
#define PFZ_HD __host__ __device__ inline
#endif
#ifndef PFZ_LDS_U16
#define PFZ_LDS_U16 uint16_t      // the kernel defines it as an LDS-address-space type: LDS and global loads must not be merged into flat ones
#endif
#ifndef PFZ_LDS_U8
#define PFZ_LDS_U8 uint8_t
#endif

namespace pfz {

enum FuzzMode { kWRatio = 0, kPartialRatio = 1, kTokenSetRatio = 2, kTokenRatio = 3, kPartialTokenSortRatio = 4,
                kPartialTokenSetRatio = 5, kPartialTokenRatio = 6 };

constexpr int kFuzzMaxTokens = 32;     // distinct tokens per string in the register / LDS kernels (token sets are 32-bit masks)
constexpr int kFuzzHistWords = 8;      // character classes: 8 dwords x 4 one-byte counters

// one from-string: match tables of its three forms, its distinct tokens
// three of a kind, one per form.  Read [v] with v known only at run time is a chain of selects -- NOT an indexed array:
// the compiler moves an indexed register array to scratch memory, and with it every later access to the struct around it
template <typename X>
struct Fz3 {
    X a0, a1, a2;
    PFZ_HD X operator[](int v) const { return v == 0 ? a0 : (v == 1 ? a1 : a2); }
    PFZ_HD void set(int v, X x)
    {
        a0 = v == 0 ? x : a0;
        a1 = v == 1 ? x : a1;
        a2 = v == 2 ? x : a2;
    }
};

template <int W> struct FuzzFrom {
    const uint64_t *pm;          // [symbol][form][W]; symbol 0 (padding / not in the to-alphabet) is all zero
    Fz3<int> la;                 // form lengths
    int ta;                      // distinct tokens (<= kFuzzMaxTokens)
    const int32_t *tid, *tlen;   // [ta]
    const uint64_t *tmask;       // [ta][W] positions of token i in form 2
    const uint64_t *smask;       // [ta][W] position of the space after token i (empty for the last token)
};

// one to-string, every array its own contiguous record: on the device the records are 16-byte aligned and padded to
// whole 8-symbol (8-tag, 4-token) chunks, so that a lane fetches 8 symbols with ONE 128-bit load
struct FuzzTo {
    Fz3<const uint16_t *> sym;   // symbols of form v
    const uint8_t *tag;          // form 2: token number (5 bits) | 0x80 for the space that follows that token
    const int32_t *tok_id, *tok_len;   // distinct tokens, in the order of form 2
    Fz3<int> lb;
    int tb;
    // optional scratch column [pos * stage_stride] of kFuzzStage symbols (LDS in the kernel, none on the host): a window sweep
    // re-reads the to-string |from| times, and a global load per recurrence step is a dependent ~500-cycle round trip
    PFZ_LDS_U16 *stage;
    int stage_stride;            // 0: there is no column
    int staged;                  // the form whose symbols the column holds (-1: none)
    int n_windows;               // (statistics) windows fz_partial swept for this pair
#ifdef PFZ_K7_PROFILE
    long long t0;                // (profiling build: tools/build_variant.sh -DPFZ_K7_PROFILE) clock at the last FZ_TICK
    unsigned int tk[6];          // ticks of this lane by sub-phase of scoring
#endif
};

// FZ_TICK(T, k): the time since the last tick goes to sub-phase k (profiling build of the kernel only)
#ifndef FZ_TICK
#define FZ_TICK(T, k)
#endif

constexpr int kFuzzStage = 64;

// eight consecutive symbols / tags (the device records are aligned and padded for it; the host reads them one by one)
PFZ_HD void fz_load8(const uint16_t *p, int n, int (&c)[8])
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 v = *(const uint4 *)p;
    c[0] = (int)(v.x & 0xffffu), c[1] = (int)(v.x >> 16), c[2] = (int)(v.y & 0xffffu), c[3] = (int)(v.y >> 16);
    c[4] = (int)(v.z & 0xffffu), c[5] = (int)(v.z >> 16), c[6] = (int)(v.w & 0xffffu), c[7] = (int)(v.w >> 16);
    (void)n;
#else
    for (int q = 0; q < 8; ++q) c[q] = q < n ? (int)p[q] : 0;
#endif
}

PFZ_HD void fz_load8(const uint8_t *p, int n, int (&c)[8])
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint2 v = *(const uint2 *)p;
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = (int)((v.x >> (8 * q)) & 0xffu), c[4 + q] = (int)((v.y >> (8 * q)) & 0xffu);
    (void)n;
#else
    for (int q = 0; q < 8; ++q) c[q] = q < n ? (int)p[q] : 0;
#endif
}

// copy form v into the scratch column; longer forms stay in global memory
PFZ_HD void fz_stage(FuzzTo &T, int v)
{
    const int lb = T.lb[v];
    if (T.stage_stride == 0 || T.staged == v || lb > kFuzzStage) return;
    for (int p0 = 0; p0 < lb; p0 += 8) {
        int c[8];
        fz_load8(T.sym[v] + p0, lb - p0, c);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (p0 + q < lb) T.stage[(p0 + q) * T.stage_stride] = (uint16_t)c[q];
    }
    T.staged = v;
}

PFZ_HD double fz_ratio_of(int lcs, int lensum)
{
    const int dist = lensum - 2 * lcs;
    const double norm_dist = lensum != 0 ? (double)dist / (double)lensum : 0.0;
    return (1.0 - norm_dist) * 100.0;
}

PFZ_HD double fz_norm_distance(int dist, int lensum)
{
    return lensum != 0 ? 100.0 - (double)(100 * dist) / (double)lensum : 100.0;
}

PFZ_HD int fz_popc64(uint64_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(v);
#else
    return __builtin_popcountll(v);
#endif
}

PFZ_HD int fz_popc32(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}

PFZ_HD int fz_last_bit(uint32_t v) { return v ? 31 - __builtin_clz(v) : -1; }
PFZ_HD int fz_min(int a, int b) { return a < b ? a : b; }
PFZ_HD int fz_max(int a, int b) { return a > b ? a : b; }
PFZ_HD double fz_fmax(double a, double b) { return a > b ? a : b; }      // (no NaNs here)

// three-input boolean functions of 64-bit words: one v_bitop3_b32 per half on the device (the compiler renders a & b & c and
// s | (v ^ u) as two two-input operations each -- four of a recurrence step's seven vector instructions per word; the kernels
// built on this step are bound by their vector instruction rate)
PFZ_HD uint64_t fz_and3(uint64_t a, uint64_t b, uint64_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0x80);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0x80);
    return (uint64_t)hi << 32 | lo;
#else
    return a & b & c;
#endif
}

// s | (v & ~u)  (= s | (v ^ u) where u is a subset of v)
PFZ_HD uint64_t fz_or_andnot(uint64_t s, uint64_t v, uint64_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)s, (uint32_t)v, (uint32_t)u, 0xF4);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(s >> 32), (uint32_t)(v >> 32), (uint32_t)(u >> 32), 0xF4);
    return (uint64_t)hi << 32 | lo;
#else
    return s | (v & ~u);
#endif
}

// (the 64-bit sum as ONE instruction: once its operands and users are 32-bit halves the compiler splits it into an add and
// an add-with-carry)
PFZ_HD uint64_t fz_add64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t d;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return a + b;
#endif
}

template <int W>
PFZ_HD void fz_step(uint64_t (&V)[W], const uint64_t *pm, const uint64_t (&mask)[W])
{
    if (W == 1) {           // (no carry to pass on)
        const uint64_t u = V[0] & pm[0] & mask[0];
        V[0] = fz_or_andnot(fz_add64(V[0], u), V[0], u);
        return;
    }
    uint64_t carry = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint64_t u = V[w] & pm[w] & mask[w];
        const uint64_t sum = V[w] + u + carry;
        carry = (sum < V[w]) | (carry & (sum == V[w]));
        V[w] = fz_or_andnot(sum, V[w], u);
    }
}

// zero bits of V in positions [0, k)
template <int W>
PFZ_HD int fz_zeros_below(const uint64_t (&V)[W], int k)
{
    int n = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int bits = fz_min(fz_max(k - 64 * w, 0), 64);
        const uint64_t m = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
        n += fz_popc64(~V[w] & m);
    }
    return n;
}

template <int W>
PFZ_HD void fz_range_mask(uint64_t (&m)[W], int lo, int hi)      // bits [lo, hi)
{
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int a = fz_min(fz_max(lo - 64 * w, 0), 64), b = fz_min(fz_max(hi - 64 * w, 0), 64);
        const uint64_t below_b = b >= 64 ? ~0ull : ((1ull << b) - 1ull), below_a = a >= 64 ? ~0ull : ((1ull << a) - 1ull);
        m[w] = below_b & ~below_a;
    }
}

// LCS state of the from-form v (restricted to `amask`) against the to-form v; with `tagged` only the to-tokens whose bit
// is set in `rb` are fed (form 2), the space after the last of them (`last_rb`) left out
#ifndef PFZ_K7_PASS_PRELOAD
#define PFZ_K7_PASS_PRELOAD 1          // (0: a branch per position, the table reads one behind the other -- A/B builds)
#endif
template <int W>
PFZ_HD void fz_lcs_pass(const FuzzFrom<W> &F, const FuzzTo &T, int v, const uint64_t (&amask)[W], bool tagged, uint32_t rb,
                        int last_rb, uint64_t (&V)[W])
{
#pragma unroll
    for (int w = 0; w < W; ++w) V[w] = ~0ull;
    const int lb = T.lb[v];
    const uint16_t *sym = T.sym[v];
    // eight positions at a time: one 128-bit load of symbols (one 64-bit load of tags), then eight recurrence steps
    for (int p0 = 0; p0 < lb; p0 += 8) {
        int sy[8], tg[8];
        fz_load8(sym + p0, lb - p0, sy);
        if (tagged) {
            fz_load8(T.tag + p0, lb - p0, tg);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = tg[q] & 31;
                const bool keep = ((rb >> j) & 1u) && !((tg[q] & 0x80) && j == last_rb);
                sy[q] = keep ? sy[q] : 0;
            }
        }
        // (no branch per position: beyond the form's end the symbol is 0, "unknown" -- its table row is empty and the step
        // leaves V as it is -- so the eight table reads go out together instead of each behind the step before it)
#if !PFZ_K7_PASS_PRELOAD
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (p0 + q < lb) fz_step<W>(V, F.pm + (sy[q] * 3 + v) * W, amask);
#else
#pragma unroll
        for (int q = 0; q < 8; ++q) sy[q] = p0 + q < lb ? sy[q] : 0;
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
            uint64_t pmv[4][W];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int w = 0; w < W; ++w) pmv[q][w] = F.pm[(sy[h + q] * 3 + v) * W + w];
#pragma unroll
            for (int q = 0; q < 4; ++q) fz_step<W>(V, pmv[q], amask);
        }
#endif
    }
}

// ---- window sweeps (rapidfuzz.fuzz.partial_ratio of two v-forms) -----------------------------------------------------------
// The windows of a pair, numbered: with the from-form the shorter (or equal) -- windows of the to-form starting at
// s = 0 .. lb - 1 (its prefixes shorter than the from-form fall out of the first window step by step); with the from-form
// the longer (or equal) -- windows of the from-form starting at i = 0 .. la - 1, one pass over the to-form each (the
// prefixes of the from-form fall out of the first pass).  Equal lengths: both families, the first then the second.
// The best window is found as an exact rational lcs / (|shorter| + |window|).
//
// Not every window is swept: moving a window by d positions brings in at most d new characters, so its LCS grows by at most
// d -- after a window with LCS l, the next windows that can neither beat the best ratio so far nor reach what the caller
// cares for are stepped over (integer arithmetic on l + d; a typical far-from-matching pair sweeps one window in
// |shorter| / 2).  The windows of a pair can be shared out ([w, w_end) below): every share is swept on its own, the best
// of the shares is the best of the pair.
PFZ_HD int fz_n_windows(int la, int lb) { return (lb >= la ? lb : 0) + (lb <= la ? la : 0); }

#ifndef FZ_ANY
#define FZ_ANY(x) (x)          // (the kernel: any lane of the wave -- a cheap way round code few lanes need)
#endif

struct FuzzSweep {
    int v, la, lb;         // the forms (both non-empty)
    int w, w_end;          // the next window, the end of this share
    int bl, bs;            // the best candidate so far: lcs, length sum
    const uint16_t *sym;   // the to-form's symbols; or, when stage_stride != 0, its copy in the scratch column
    PFZ_LDS_U16 *stage;
    int stage_stride;      // (in elements)
    bool narrow;           // the column holds bytes (an alphabet of at most 255 symbols: half the scratch memory)
    // (an experiment of tests/k7_core_host.cpp, not in the kernel: forms of at most 64 symbols) bit p of live_to: the to-form's
    // symbol at p occurs in the from-form; bit q of live_from: the from-form's symbol at q occurs in the to-form.  A window
    // that moves on gains matches only from LIVE positions that enter it.
    bool has_live;
    uint64_t live_to, live_from;
    int n_swept;           // (LIVE) windows whose recurrence was run: the others were settled by their live positions
};

// copy a to-form of at most kFuzzStage symbols into the scratch column (a window sweep re-reads it |from| times)
PFZ_HD void fz_stage_form(const uint16_t *sym, int lb, PFZ_LDS_U16 *stage, int stride, bool narrow)
{
    for (int p0 = 0; p0 < lb; p0 += 8) {
        int c[8];
        fz_load8(sym + p0, lb - p0, c);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (p0 + q < lb) {
                if (narrow) ((PFZ_LDS_U8 *)stage)[(p0 + q) * stride] = (uint8_t)c[q];
                else stage[(p0 + q) * stride] = (uint16_t)c[q];
            }
    }
}

// (sym / stage: see FuzzSweep; stage_stride 0 = read the symbols where they are)
PFZ_HD void fz_sweep_begin(FuzzSweep &S, int v, int la, int lb, int w, int w_end, const uint16_t *sym, PFZ_LDS_U16 *stage,
                           int stage_stride, bool narrow = false)
{
    S.narrow = narrow;
    S.has_live = false;
    S.live_to = S.live_from = 0ull;
    S.n_swept = 0;
    S.v = v;
    S.la = la;
    S.lb = lb;
    S.w = w;
    S.w_end = w_end;
    S.bl = 0;
    S.bs = 1;
    S.sym = sym;
    S.stage = stage;
    S.stage_stride = stage_stride;
}

PFZ_HD int fz_sweep_sym(const FuzzSweep &S, int pos)
{
    if (S.stage_stride == 0) return (int)S.sym[pos];
    return S.narrow ? (int)((PFZ_LDS_U8 *)S.stage)[pos * S.stage_stride] : (int)S.stage[pos * S.stage_stride];
}

// sweeps window S.w and moves S.w to the next window that can matter; true: the share is done.  A window matters when it
// can beat the best so far AND reach `thr` after the factor f the caller's formula multiplies partial_ratio by
// (200 lcs f / sum >= thr; thr is the caller's floor minus a margin far above the rounding of these products).
// LIVE (an experiment of the CPU harness, not instantiated by the kernel): use S.live_to / live_from -- see FuzzSweep.
template <int W, bool LIVE = false>
PFZ_HD bool fz_sweep_window(FuzzSweep &S, const FuzzFrom<W> &F, double f, double thr)
{
    const int la = S.la, lb = S.lb, v = S.v;
    const int n_first = lb >= la ? lb : 0;
    const bool of_to = S.w < n_first;                           // a window of the to-form (else: of the from-form)
    const int idx = of_to ? S.w : S.w - n_first;
    const int lm = fz_min(la, lb), ll = fz_max(la, lb);
    const int wlen = fz_min(lm, ll - idx);
    const int t0 = of_to ? idx : 0, t1 = of_to ? idx + wlen : lb;       // to-positions fed
    const int c_lo = of_to ? 0 : idx, c_hi = of_to ? la : idx + wlen;   // from-positions allowed (from c_lo on) and counted
    int bl = S.bl, bs = S.bs;
    auto cand = [&](int lcs, int sum) {
        if (lcs * bs > bl * sum) {          // (lengths <= 256 W... <= 1024: the products fit an int)
            bl = lcs;
            bs = sum;
        }
    };
    auto worth = [&](int lcs, int sum) { return lcs * bs > bl * sum && !(200.0 * (double)lcs * f < thr * (double)sum); };
    uint64_t m[W], V[W];
    fz_range_mask<W>(m, c_lo, la);
#pragma unroll
    for (int w = 0; w < W; ++w) V[w] = ~0ull;
    const bool prefixes = of_to && idx == 0;
    // (live masks: a window matches at most its live positions -- and any part of it at least as many characters as it
    // matches; when that cannot matter, the window is not swept and the count stands in for its LCS in the step below)
    auto live_in = [&](int lo, int len) {
        const uint64_t live = of_to ? S.live_to : S.live_from;
        return fz_popc64((live >> lo) & (len >= 64 ? ~0ull : ((1ull << len) - 1ull)));
    };
    int l_known = -1;
    if constexpr (LIVE) {
        const int c = fz_min(live_in(idx, wlen), wlen);
        if (!worth(c, lm + (idx == 0 ? c : wlen))) l_known = c;
        S.n_swept += l_known < 0;
    }
    for (int k = t0; k < t1 && (!LIVE || l_known < 0); k += 4) {
        int sy[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sy[q] = k + q < t1 ? fz_sweep_sym(S, k + q) : 0;
        uint64_t pmv[4][W];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int w = 0; w < W; ++w) pmv[q][w] = F.pm[(sy[q] * 3 + v) * W + w];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fz_step<W>(V, pmv[q], m);
            // a prefix of the to-form, k + q + 1 characters long (shorter than the from-form): only the long ones can matter
            const int fed = k + q + 1;
            const bool look = prefixes && fed < la && fed <= t1 && worth(fed, la + fed);
            if (FZ_ANY(look))
                if (look) cand(fz_zeros_below<W>(V, la), la + fed);
        }
    }
    const int l = LIVE && l_known >= 0 ? l_known : fz_zeros_below<W>(V, c_hi) - fz_zeros_below<W>(V, c_lo);
    if (!LIVE || l_known < 0) {
        cand(l, lm + wlen);
        if (!of_to && idx == 0)
            for (int k = 1; k < lb; ++k) cand(fz_zeros_below<W>(V, k), lb + k);          // prefixes of the from-form
    }
    // the next window that can matter.  Whole windows (starts up to ll - lm) all need the same LCS: the step over them is
    // one subtraction; the shrinking windows at the end are tried one by one
    int d = 1;
    const int last_whole = ll - lm;
    if constexpr (LIVE) {
        // the window d positions on holds at most the matches of this one plus the live positions that entered
        const uint64_t live = of_to ? S.live_to : S.live_from;
        int entered = 0;
        while (idx + d < ll) {
            const int wl = fz_min(lm, ll - idx - d), enter = idx + wlen + d - 1;
            entered += enter < ll ? (int)((live >> enter) & 1ull) : 0;
            if (worth(fz_min(fz_min(l + entered, live_in(idx + d, wl)), wl), lm + wl)) break;
            ++d;
        }
    }
    if (!LIVE && idx < last_whole) {
        const int sum = 2 * lm;
        int need = bl * sum / bs + 1;                        // ... to beat the best so far
#if defined(__HIP_DEVICE_COMPILE__)
        int reach = (int)((float)thr * (float)sum * __builtin_amdgcn_rcpf(200.0f * (float)f)) - 1;      // ... to reach thr, nearly:
#else
        int reach = (int)((float)thr * (float)sum / (200.0f * (float)f)) - 1;
#endif
        reach = fz_max(reach, 0);
        while (200.0 * (double)reach * f < thr * (double)sum) ++reach;       // ... exactly (a step or two)
        need = fz_max(need, reach);
        // (a whole window holds at most lm matches: beyond that none of them matters; else on to the first that may, or to
        // the first of the shrinking windows)
        d = need > lm ? last_whole + 1 - idx : fz_min(fz_max(need - l, 1), last_whole + 1 - idx);
    }
    if (!LIVE && idx + d > last_whole)
        while (idx + d < ll) {
            const int wl = fz_min(lm, ll - idx - d);
            if (worth(fz_min(l + d, wl), lm + wl)) break;
            ++d;
        }
    S.bl = bl;
    S.bs = bs;
    S.w += d;              // (stops at the end of its family: the other family starts with a window of its own)
    return S.w >= S.w_end;
}

// partial_ratio of the two v-forms, every window in one share.  `floor`: ratios below it do not matter to the caller
// (0: all do); the result is exact whenever it is >= floor.
template <int W>
PFZ_HD double fz_partial(const FuzzFrom<W> &F, FuzzTo &T, int v, double floor)
{
    const int la = F.la[v], lb = T.lb[v];
    if (la == 0 || lb == 0) return la == 0 && lb == 0 ? 100.0 : 0.0;
    fz_stage(T, v);
    FuzzSweep S;
    const bool staged = T.staged == v;
    fz_sweep_begin(S, v, la, lb, 0, fz_n_windows(la, lb), T.sym[v], T.stage, staged ? T.stage_stride : 0);
    while (!fz_sweep_window<W>(S, F, 1.0, floor - 1e-7)) T.n_windows += 1;
    T.n_windows += 1;
    return fz_ratio_of(S.bl, S.bs);
}

// what a swept partial_ratio p of the v-forms contributes to the pair's score under `mode` -- the factors in the order
// fz_score's formulas apply them -- and (fz_sweep_factor) their product, for fz_sweep_window's test
PFZ_HD double fz_sweep_score(int mode, int v, int la0, int lb0, double p)
{
    if (mode != kWRatio) return p;
    const double scale = fz_max(la0, lb0) < 8 * fz_min(la0, lb0) ? 0.9 : 0.6;
    return v == 0 ? p * scale : p * 0.95 * scale;
}

PFZ_HD double fz_sweep_factor(int mode, int v, int la0, int lb0)
{
    if (mode != kWRatio) return 1.0;
    const double scale = fz_max(la0, lb0) < 8 * fz_min(la0, lb0) ? 0.9 : 0.6;
    return v == 0 ? scale : 0.95 * scale;
}

// common distinct tokens: bit i of ca (from-tokens), bit j of cb (to-tokens)
template <int W>
PFZ_HD void fz_intersect(const FuzzFrom<W> &F, const FuzzTo &T, uint32_t &ca, uint32_t &cb)
{
    ca = cb = 0u;
    for (int j = 0; j < T.tb; ++j) {
        const int idb = T.tok_id[j];
        for (int i = 0; i < F.ta; ++i)
            if (F.tid[i] == idb) {
                ca |= 1u << i;
                cb |= 1u << j;
            }
    }
}

template <int W>
PFZ_HD double fz_token_set(const FuzzFrom<W> &F, const FuzzTo &T, uint32_t ca, uint32_t cb)
{
    const int ta = F.ta, tb = T.tb;
    if (ta == 0 || tb == 0) return 0.0;
    const int nc = fz_popc32(ca);
    if (nc > 0 && (nc == ta || nc == tb)) return 100.0;
    // lengths of the joined differences and of the joined intersection
    const uint32_t ra = ~ca & (ta >= 32 ? ~0u : ((1u << ta) - 1u)), rb = ~cb & (tb >= 32 ? ~0u : ((1u << tb) - 1u));
    int ab_len = fz_popc32(ra) - 1, ba_len = fz_popc32(rb) - 1, sect_len = nc > 0 ? nc - 1 : 0;
    uint64_t amask[W];
#pragma unroll
    for (int w = 0; w < W; ++w) amask[w] = 0ull;
    const int last_ra = fz_last_bit(ra), last_rb = fz_last_bit(rb);
    for (int i = 0; i < ta; ++i) {
        const bool rem = (ra >> i) & 1u;
        ab_len += rem ? F.tlen[i] : 0;
        sect_len += rem ? 0 : F.tlen[i];
#pragma unroll
        for (int w = 0; w < W; ++w) amask[w] |= rem ? (F.tmask[i * W + w] | (i != last_ra ? F.smask[i * W + w] : 0ull)) : 0ull;
    }
    for (int j = 0; j < tb; ++j)
        if ((rb >> j) & 1u) ba_len += T.tok_len[j];
    uint64_t V[W];
    fz_lcs_pass<W>(F, T, 2, amask, true, rb, last_rb, V);
    const int lcs = fz_zeros_below<W>(V, F.la[2]);
    const int sect_sep = sect_len != 0 ? 1 : 0;
    const int sect_ab_len = sect_len + sect_sep + ab_len, sect_ba_len = sect_len + sect_sep + ba_len;
    const double result = fz_norm_distance(ab_len + ba_len - 2 * lcs, sect_ab_len + sect_ba_len);
    if (sect_len == 0) return result;
    const double r_ab = fz_norm_distance(sect_sep + ab_len, sect_len + sect_ab_len);
    const double r_ba = fz_norm_distance(sect_sep + ba_len, sect_len + sect_ba_len);
    return fz_fmax(result, fz_fmax(r_ab, r_ba));
}

// The score of the pair under `mode`, exact whenever it is >= cur (below cur it may come out lower than the true score,
// never higher: components that cannot reach cur are left out).
// Every mode is a combination of three kinds of work -- plain LCS passes of a form (bit v of need_l), the masked
// token-set pass, window sweeps of a form (bit v of want_p, decided after the passes: a window has at most the LCS of the
// whole forms and at least that many characters, so ratio_of(lcs, |shorter| + lcs) bounds partial_ratio) -- laid out so
// that each kind has ONE call site: the kernel inlines one copy of each, whatever the mode.
// DEFER: the window sweeps are left to the caller -- *want gets the forms to sweep (bit v), the returned score is that of
// the other components (the pair's score is the maximum of it and fz_sweep_score() of every swept form).
template <int W, bool DEFER = false>
PFZ_HD double fz_score(const FuzzFrom<W> &F, FuzzTo &T, int mode, double cur, int *want = nullptr)
{
    if (DEFER) *want = 0;
    const int la0 = F.la[0], lb0 = T.lb[0], ta = F.ta, tb = T.tb;
    const bool toks = ta != 0 && tb != 0;
    uint32_t ca = 0u, cb = 0u;
    if (mode != kPartialRatio && mode != kPartialTokenSortRatio) fz_intersect<W>(F, T, ca, cb);
    bool near = false;
    double scale = 1.0;
    int need_l = 0, sweep_of = 0;           // sweep_of: forms a sweep may follow the pass of (staged before the pass)
    bool need_ts = false;
    if (mode == kWRatio) {
        if (la0 == 0 || lb0 == 0) return 0.0;
        const int lmax = fz_max(la0, lb0), lmin = fz_min(la0, lb0);
        near = 2 * lmax < 3 * lmin;                               // len_ratio < 1.5
        scale = lmax < 8 * lmin ? 0.9 : 0.6;                     // len_ratio < 8
        need_l = 1;
        if (near) {
            if (!(100.0 * 0.95 < cur)) {                          // the token scorers are <= 100
                need_l |= 2;
                need_ts = true;
            }
        }
        else {
            sweep_of = 1;
            // the distinct-token form is a subsequence of the sorted-token form: one pass (form 1) bounds both sweeps
            if (toks && !ca && !(100.0 * 0.95 * scale < cur)) need_l |= 2, sweep_of |= 2;
        }
    }
    else if (mode == kPartialRatio) need_l = sweep_of = 1;
    else if (mode == kTokenSetRatio) need_ts = true;
    else if (mode == kTokenRatio) need_l = 2, need_ts = true;
    else if (mode == kPartialTokenSortRatio) need_l = sweep_of = 2;
    else if (mode == kPartialTokenSetRatio) need_l = sweep_of = (toks && !ca) ? 4 : 0;
    else need_l = sweep_of = (toks && !ca) ? 6 : 0;

    Fz3<int> lcs = {0, 0, 0};
    uint64_t all[W], V[W];
#pragma unroll
    for (int w = 0; w < W; ++w) all[w] = ~0ull;
    FZ_TICK(T, 0);
    for (int v = 0; v < 3; ++v)
        if ((need_l >> v) & 1) {
            if ((sweep_of >> v) & 1) fz_stage(T, v);
            fz_lcs_pass<W>(F, T, v, all, false, 0u, 0, V);
            lcs.set(v, fz_zeros_below<W>(V, F.la[v]));
        }
    FZ_TICK(T, 1);
    const double tset = need_ts ? fz_token_set<W>(F, T, ca, cb) : 0.0;
    FZ_TICK(T, 2);

    // window sweeps: p[v] = partial_ratio of the v-forms, or 0 (a lower bound) where it cannot reach cur / factor
    Fz3<double> p = {0.0, 0.0, 0.0}, pfac = {1.0, 1.0, 1.0};
    int want_p = 0;
    // (f1, f2: the factors the caller's formula multiplies p[v] by, in its order -- the test below must round as it does)
    auto consider = [&](int v, int l, double f1, double f2) {    // l: an upper bound of the LCS of any two windows of the v-forms
        const int la = F.la[v], lb = T.lb[v];
        if (la == 0 || lb == 0) p.set(v, la == 0 && lb == 0 ? 100.0 : 0.0);
        else {
            const int lm = fz_min(la, lb), c = fz_min(l, lm);
            if (!(fz_ratio_of(c, lm + c) * f1 * f2 < cur)) {
                want_p |= 1 << v;
                if (!DEFER) pfac.set(v, f1 * f2);
            }
        }
    };
    if (mode == kWRatio) {
        if (!near) {
            consider(0, lcs[0], scale, 1.0);
            if (need_l & 2) {
                consider(1, lcs[1], 0.95, scale);
                consider(2, lcs[1], 0.95, scale);
            }
        }
    }
    else if (mode == kPartialRatio) consider(0, lcs[0], 1.0, 1.0);
    else if (mode == kPartialTokenSortRatio) consider(1, lcs[1], 1.0, 1.0);
    else if (mode == kPartialTokenSetRatio) {
        if (need_l) consider(2, lcs[2], 1.0, 1.0);
    }
    else if (mode == kPartialTokenRatio && need_l) {
        consider(1, lcs[1], 1.0, 1.0);
        consider(2, lcs[2], 1.0, 1.0);
    }
    if (DEFER) *want = want_p;
    else
        for (int v = 0; v < 3; ++v)
            if ((want_p >> v) & 1) p.set(v, fz_partial<W>(F, T, v, cur / pfac[v]));
    FZ_TICK(T, 3);

    switch (mode) {
    case kWRatio: {
        const double end_ratio = fz_ratio_of(lcs[0], la0 + lb0);
        if (near) {
            if (!need_ts) return end_ratio;
            // (no tokens on either side: ratio("", "") = 100, as rapidfuzz)
            return fz_fmax(end_ratio, fz_fmax(fz_ratio_of(lcs[1], F.la[1] + T.lb[1]), tset) * 0.95);
        }
        const double pt = !toks ? 0.0 : (ca ? 100.0 : fz_fmax(p[1], p[2]));
        return fz_fmax(fz_fmax(end_ratio, p[0] * scale), pt * 0.95 * scale);
    }
    case kPartialRatio: return p[0];
    case kTokenSetRatio: return tset;
    case kTokenRatio: return fz_fmax(fz_ratio_of(lcs[1], F.la[1] + T.lb[1]), tset);
    case kPartialTokenSortRatio: return p[1];
    case kPartialTokenSetRatio: return !toks ? 0.0 : (ca ? 100.0 : p[2]);
    default: return !toks ? 0.0 : (ca ? 100.0 : fz_fmax(p[1], p[2]));
    }
}

// ---- upper bound -------------------------------------------------------------------------------------------------------

// what the bound needs of one string: form lengths, distinct tokens, the character-class histogram (one byte per class,
// four classes per dword) with its sum -- usum < 0: no histogram (a class count beyond 255) -- and a 64-bit signature
// of the token ids (bit = hash of the id: disjoint signatures <=> certainly no common token)
struct FuzzSummary {
    int len[3], ntok;
    uint32_t hist[kFuzzHistWords];
    int usum;
    uint64_t sig;
};

PFZ_HD uint64_t fz_sig_bit(int32_t tok_id) { return 1ull << (((uint32_t)tok_id * 0x9E3779B1u) >> 26); }

PFZ_HD int fz_sad_u8(uint32_t a, uint32_t b, int acc)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)__builtin_amdgcn_sad_u8(a, b, (uint32_t)acc);
#else
    for (int k = 0; k < 4; ++k) {
        const int x = (a >> (8 * k)) & 255, y = (b >> (8 * k)) & 255;
        acc += x > y ? x - y : y - x;
    }
    return acc;
#endif
}

// sum over the character classes of min(count_a, count_b): >= the LCS of any two sub-multisets of the strings' characters
PFZ_HD int fz_common_chars(const FuzzSummary &a, const FuzzSummary &b)
{
    if (a.usum < 0 || b.usum < 0) return fz_min(a.len[0], b.len[0]);      // (no form is longer than the string itself)
    int sad = 0;
#pragma unroll
    for (int d = 0; d < kFuzzHistWords; ++d) sad = fz_sad_u8(a.hist[d], b.hist[d], sad);
    return (a.usum + b.usum - sad) >> 1;
}

// Symbol presence (NOT YET IN THE KERNEL: measured on the CPU, tests/test_k7_core_cpu.py::test_symbol_presence_bound_* -- 30 %
// fewer pairs reach a row's best score on config 3's titles, with 64 bits as well as with 256; the device side needs 8 B
// more per to-string summary and ~16 vector instructions per pair in sweep 1).
// pres: one bit per symbol rank (mod 64), the space left out -- joined forms gain and lose spaces.  Folding ranks onto
// shared bits only merges symbols: a bit absent on the other side still means every symbol on it is absent there.
constexpr int kFuzzPresWords = 2;
PFZ_HD void fz_presence_miss(const uint32_t *pa, const uint32_t *pb, int &miss_a, int &miss_b)
{
    int na = 0, nb = 0, common = 0;
#pragma unroll
    for (int w = 0; w < kFuzzPresWords; ++w) {
        na += fz_popc32(pa[w]);
        nb += fz_popc32(pb[w]);
        common += fz_popc32(pa[w] & pb[w]);
    }
    miss_a = na - common;
    miss_b = nb - common;
}

// 200 lcs / lensum with the hardware reciprocal (1 ulp; the caller's slack is five orders of magnitude wider): a correctly
// rounded float division is ten instructions, and the bound of one pair holds up to seven of them
PFZ_HD float fz_r32(int lcs, int lensum)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return lensum > 0 ? 200.0f * (float)lcs * __builtin_amdgcn_rcpf((float)lensum) : 100.0f;
#else
    return lensum > 0 ? 200.0f * (float)lcs / (float)lensum : 100.0f;
#endif
}

// (the same where the caller knows lensum > 0)
PFZ_HD float fz_r32_nz(int lcs, int lensum)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 200.0f * (float)lcs * __builtin_amdgcn_rcpf((float)lensum);
#else
    return 200.0f * (float)lcs / (float)lensum;
#endif
}

// token_set_ratio's bound once the common tokens of the from-string are known (bit i of ca; lb2 / tb: the to-string's
// form-2 length and distinct tokens): the two "sect" ratios are length arithmetic -- exact -- and the LCS of the joined
// differences is at most the shorter difference and at most u minus the characters of the common tokens (they sit in both
// histograms and in neither difference).  Only from-side token lengths are needed: a form 2 is its tokens joined by
// single spaces, so the to-side difference is what is left of its length.
// (nc common tokens with sect_chars characters between them; la2 / ta, lb2 / tb: form-2 length and distinct tokens of
// the two strings.  Selects only: the sweep runs it for all 64 lanes.)
// (miss_a / miss_b: distinct symbols of the one string that the other lacks -- see fz_presence_miss; they all sit in the
// token differences: a common token has every symbol on both sides)
PFZ_HD float fz_token_set_bound_n(int la2, int ta, int lb2, int tb, int u, int nc, int sect_chars, int miss_a = 0, int miss_b = 0)
{
    // joined length of k tokens with c characters in total: c + k - 1
    const int ab_len = (la2 - (ta - 1) - sect_chars) + (ta - nc) - 1, ba_len = (lb2 - (tb - 1) - sect_chars) + (tb - nc) - 1;
    const int sect_len = sect_chars + (nc > 0 ? nc - 1 : 0), sect_sep = sect_len != 0 ? 1 : 0;
    const int sect_ab_len = sect_len + sect_sep + ab_len, sect_ba_len = sect_len + sect_sep + ba_len;
    const int m = fz_min(fz_min(ab_len - miss_a, ba_len - miss_b), fz_max(u - sect_chars, 0));
    // (100 - 100 d / s = 100 (s - d) / s: the same expression shape as fz_r32, hardware reciprocal included)
    const int total = sect_ab_len + sect_ba_len;
    const float result = total > 0 ? 0.5f * fz_r32(total - (ab_len + ba_len - 2 * m), total) : 100.0f;
    const float r_ab = 0.5f * fz_r32(sect_len + sect_ab_len - (sect_sep + ab_len), sect_len + sect_ab_len);
    const float r_ba = 0.5f * fz_r32(sect_len + sect_ba_len - (sect_sep + ba_len), sect_len + sect_ba_len);
    const float sect_best = r_ab > r_ba ? r_ab : r_ba;
    const float r = sect_len == 0 ? result : (result > sect_best ? result : sect_best);
    const float full = (nc > 0 && (nc == ta || nc == tb)) ? 100.0f : r;       // one token set inside the other
    return (ta == 0 || tb == 0) ? 0.0f : full;
}

template <int W>
PFZ_HD float fz_token_set_bound(const FuzzFrom<W> &F, uint32_t ca, int lb2, int tb, int u, int miss_a = 0, int miss_b = 0)
{
    int sect_chars = 0;
    for (int i = 0; i < F.ta; ++i) sect_chars += ((ca >> i) & 1u) ? F.tlen[i] : 0;
    return fz_token_set_bound_n(F.la[2], F.ta, lb2, tb, u, fz_popc32(ca), sect_chars, miss_a, miss_b);
}

// An upper bound of fz_score(F, T, mode, .) -- of the TRUE score -- from the two summaries alone.  `common` says what is
// known about common tokens: -1 unknown (the signatures intersect; assume the best), 0 none, 1 some -- then `tset` is
// fz_token_set_bound (pass a negative value to assume 100).  float32: the caller keeps a slack (prune only when
// bound + 0.05 < cur).
//
// miss_a / miss_b (fz_presence_miss; 0 = not known): distinct non-space symbols of the one string that the other lacks.
// Every form of a string holds every one of its symbols at least once, and a position holding a symbol the other string
// lacks matches nothing: LCS(form of a, anything of b) <= |form of a| - miss_a, and the same with the roles swapped; a
// window sweep aligns the WHOLE shorter form, so its misses count (equal lengths: both families run, the smaller counts).
PFZ_HD float fz_upper_bound(const FuzzSummary &a, const FuzzSummary &b, int mode, int u, int common, float tset = -1.0f, int miss_a = 0,
                            int miss_b = 0)
{
    // (written with selects, not branches: the lanes of a wave take every path of this function between them)
    auto fmx = [](float x, float y) { return x > y ? x : y; };
    auto ratio_ub = [&](int v) { return fz_r32(fz_min(u, fz_min(a.len[v] - miss_a, b.len[v] - miss_b)), a.len[v] + b.len[v]); };
    auto partial_ub = [&](int v) -> float {
        const int la = a.len[v], lb = b.len[v];
        const int miss = la < lb ? miss_a : (lb < la ? miss_b : fz_min(miss_a, miss_b));
        const int mn = fz_min(la, lb), m = fz_min(u, mn - miss);
        const float r = fz_r32(m, mn + m);                       // a window has at most m matches and at least m characters
        return mn != 0 ? r : ((a.len[v] | b.len[v]) == 0 ? 100.0f : 0.0f);
    };
    const bool toks = a.ntok != 0 && b.ntok != 0;
    // token_set_ratio: 100 is possible as soon as there may be a common token; without one it is the (other
    // normalisation of the) ratio of the distinct-token forms
    auto token_set_ub = [&]() -> float {
        const float with_common = tset >= 0.0f ? tset : 100.0f, t = common != 0 ? with_common : ratio_ub(2);
        return toks ? t : 0.0f;
    };
    auto ptoken_ub = [&]() -> float {
        const float t = common != 0 ? 100.0f : fmx(partial_ub(1), partial_ub(2));
        return toks ? t : 0.0f;
    };
    switch (mode) {
    case kWRatio: {
        // Sweep 1 runs this for every pair and the kernel is bound by its vector instruction rate, so what cannot change the
        // result is left out: (i) WRatio is 0 when either string is empty, so form 0's length sum and minimum are not 0 where
        // the result counts -- its fz_r32 / partial_ub guards go (lanes with an empty string compute garbage that the last
        // select discards; forms 1 and 2 keep theirs: a string of spaces has no token); (ii) the sorted form (1) and the
        // distinct-token form (2) of a title are, nearly always, as long as the title itself: a ratio or partial bound over
        // forms of the SAME lengths as form 0 equals form 0's, enters scaled by 0.95 and is dominated by it -- computed
        // only where a length differs (a divergent branch: whole waves skip it).
        const int la = a.len[0], lb = b.len[0];
        const int lmax = fz_max(la, lb), lmin = fz_min(la, lb);
        auto r_ub = [&](int v) { return v == 0 ? fz_r32_nz(fz_min(u, fz_min(la - miss_a, lb - miss_b)), la + lb) : ratio_ub(v); };
        auto p_ub = [&](int v) -> float {
            if (v != 0) return partial_ub(v);
            const int miss = la < lb ? miss_a : (lb < la ? miss_b : fz_min(miss_a, miss_b));
            const int m = fz_min(u, lmin - miss);
            return fz_r32_nz(m, lmin + m);
        };
        const bool same1 = a.len[1] == la && b.len[1] == lb, same2 = a.len[2] == la && b.len[2] == lb;
        const float r0 = r_ub(0);
        float ub;
        if (2 * lmax < 3 * lmin) {                       // (groups are sorted by length: mostly one side per wave)
            float r1 = 0.0f, ts = 0.0f;
            if (!same1) r1 = r_ub(1);
            if (toks) {
                if (common != 0) ts = tset >= 0.0f ? tset : 100.0f;
                else if (!same2) ts = r_ub(2);
            }
            ub = fmx(r0, 0.95f * fmx(r1, ts));
        } else {
            const float scale = lmax < 8 * lmin ? 0.9f : 0.6f;
            float pt = 0.0f;
            if (toks) {
                if (common != 0) pt = 100.0f;
                else {
                    if (!same1) pt = p_ub(1);
                    if (!same2) pt = fmx(pt, p_ub(2));
                }
            }
            ub = fmx(r0, fmx(scale * p_ub(0), 0.95f * scale * pt));
        }
        return lmin != 0 ? ub : 0.0f;
    }
    case kPartialRatio: return partial_ub(0);
    case kTokenSetRatio: return token_set_ub();
    case kTokenRatio: return fmx(ratio_ub(1), token_set_ub());
    case kPartialTokenSortRatio: return partial_ub(1);
    case kPartialTokenSetRatio: {
        const float t = common != 0 ? 100.0f : partial_ub(2);
        return toks ? t : 0.0f;
    }
    default: return ptoken_ub();
    }
}

}  // namespace pfz
