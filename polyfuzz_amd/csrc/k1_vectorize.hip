// K1/K2 -- character-n-gram TF-IDF vectorisation on the device.
//
// Replaces TfidfVectorizer(min_df=1, analyzer=TFIDF._create_ngrams).fit /
// .transform as used by reference polyfuzz/models/_tfidf.py:102-118, with the
// analyzer of _tfidf.py:120-146 (clean -> sliding n-char windows, windows
// containing ' ' dropped).  sklearn semantics restated (feature_extraction/
// text.py): vocabulary = distinct n-grams in sorted order (column id = rank,
// :1194-1199), tf = raw count (:1247-1310), out-of-vocabulary n-grams ignored at
// transform (:1271-1273), idf = ln((1+n_docs)/(1+df))+1 (:1662-1679), value =
// tf*idf, rows L2-normalised sequentially in float64 (:1716-1722).
//
// Device representation
//   n-gram code : the n characters, each mapped to a small rank (1..S, 0 = pad)
//                 of w bits, packed big-endian and left-aligned into
//                 ngram_hi*w <= 64 bits.  Numeric order of codes == Python's
//                 lexicographic order of the n-gram strings (shorter prefix
//                 first), so "rank of the code among the distinct codes" is
//                 exactly sklearn's column id.
//   vocabulary  : up to 32-bit codes: presence bitmap over the code space (32 KiB
//                 for cleaned 3-grams, 512 MiB at 32 bits) + an int32
//                 rank prefix per 256-bit group: column id = prefix[g] +
//                 popcount of the lower bits -- no sort, no hash, no search.
//                 Wider codes (up to 64 bits): the distinct codes as a sorted array
//                 (gather + bitonic sort + unique), column id by binary search.
//   per string  : a slot range of len*R uint64 in HBM (R = number of n values),
//                 holding first the codes, then in place the sorted distinct
//                 (column id, tf) pairs.
// Kernels: k_extract (thread per string: clean + window + emit + mark bitmap),
// k_group_popc (+ scan), k_rows_short (wave per string: 64-lane bitonic sort,
// ballot run-lengths, df atomics), k_rows_long (workgroup per long string),
// k_idf, k_finalize (16 lanes per string: fp64 tf*idf, sum of squares replayed in index order, CSR).
// All of it is HBM-streaming work over a few MB; none of it is MFMA-shaped.
#include "pfz_internal.h"

#include <string.h>

#include <algorithm>
#include <atomic>
#include <type_traits>

namespace pfz {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;
constexpr int kShortMax = 128;     // n-grams per string k_rows_short sorts in a wave's registers (one or two keys per lane)
constexpr int kLongMax = 4096;     // n-grams per string sorted in LDS by k_rows_long
constexpr int kLongTile = 16;      // strings per scheduling tile of k_rows_long
constexpr int kMaxCodeBits = 64;     // n-gram codes are uint64
constexpr int kBitmapMaxBits = 32;   // presence bitmap (512 MiB at 32 bits) up to here, sorted code array beyond

static std::atomic<uint64_t> g_gen{1};

struct VocabView {
    const uint32_t *bitmap;
    const int32_t *prefix;
    const uint64_t *vcodes;   // sorted-vocabulary mode (codes wider than kBitmapMaxBits bits): the distinct codes,
    int64_t vocab;            // ascending; column id = position, found by binary search
};

// column id of a code, or kInvalid when the code is not in the vocabulary
__device__ inline uint32_t vocab_rank(const VocabView &v, uint64_t code)
{
    if (v.vcodes) {
        int64_t lo = 0, hi = v.vocab;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (v.vcodes[mid] < code) lo = mid + 1;
            else hi = mid;
        }
        return lo < v.vocab && v.vcodes[lo] == code ? (uint32_t)lo : kInvalid;
    }
    const uint64_t g = code >> 8;
    const uint32_t wi = (uint32_t)(code >> 5) & 7u;
    const uint32_t bit = (uint32_t)code & 31u;
    // the whole 256-bit group in two 16-byte loads (independent of each other and of the prefix load)
    const uint4 *w4 = (const uint4 *)(v.bitmap + g * 8);
    const uint4 lo = w4[0], hi = w4[1];
    const uint32_t words[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t r = (uint32_t)v.prefix[g];
    uint32_t present = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        const uint32_t m = j < wi ? 0xFFFFFFFFu : (j == wi ? ((1u << bit) - 1u) : 0u);
        r += __popc(words[j] & m);
        present |= j == wi ? (words[j] >> bit) & 1u : 0u;
    }
    return present ? r : kInvalid;
}

struct ExtractParams {
    int32_t lo, hi, clean, remove_space, w;
    int32_t bitmap_words;   // words of the vocabulary bitmap (only read by the LDS-bitmap variant)
    int64_t alpha_len;
};

// A second list riding in the same launch (a fit over to + from: both lists' extraction / rows / histograms are short,
// latency-bound launches -- one launch over both instead of two in a row; workgroups from grid0 on belong to it).
struct ListB {
    const void *chars = nullptr;
    const int64_t *off = nullptr;
    int64_t n = 0;
    uint64_t *slots = nullptr;
    int32_t *row_cnt = nullptr;       // (row_nnz follows it: row_cnt + n + 1)
    unsigned grid0 = 0xffffffffu;     // workgroups of the first list (0xffffffff: there is no second one)
};

constexpr int kLdsBitmapWords = 8192;   // code spaces of up to 18 bits (cleaned 3-grams) fit a 32 KiB LDS bitmap

// ---------------------------------------------------------------------------
// k_extract: one thread per string.
// ---------------------------------------------------------------------------
// LB: the workgroup first marks its n-grams in an LDS copy of the bitmap and merges the non-zero
// words into the global one at the end -- a probe of the global bitmap inside the per-character loop
// is a dependent L2 round trip per n-gram, and the loop of the longest string is the kernel's runtime.
// CODE: uint32_t when the n-gram codes fit 32 bits (cleaned text up to 5-grams): half the ALU work of the
// per-character loop, which is what this kernel's run time consists of.
template <int CW, bool LB, typename CODE>
__global__ __launch_bounds__(256) void k_extract(const void *__restrict__ chars_v, const int64_t *__restrict__ off,
                                                  int64_t n, ExtractParams P, const uint32_t *__restrict__ alpha_map,
                                                  uint64_t *__restrict__ slots, int32_t *__restrict__ row_cnt,
                                                  uint32_t *__restrict__ bitmap)
{
    // The 256 strings of a workgroup are contiguous in the packed buffer: stage their
    // characters into LDS with coalesced 4-byte loads (a per-thread byte walk over
    // global memory costs one uncoalesced load per character), then walk from LDS.
    constexpr int kStageBytes = 24 * 1024;
    __shared__ uint32_t stage[kStageBytes / 4];
    __shared__ uint32_t lbm[LB ? kLdsBitmapWords : 1];
    if (LB)
        for (int t = threadIdx.x; t < P.bitmap_words; t += 256) lbm[t] = 0u;
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int64_t i_end = i0 + 256 < n ? i0 + 256 : n;
    const int64_t byte0 = off[i0] * CW, byte1 = off[i_end] * CW;
    const int64_t base4 = byte0 & ~(int64_t)3;                 // 4-byte aligned start
    const bool staged = byte1 - base4 <= kStageBytes;
    if (staged) {
        const uint32_t *src = (const uint32_t *)((const uint8_t *)chars_v + base4);
        const int words = (int)((byte1 - base4 + 3) >> 2);     // the buffer is padded by 16 bytes
        for (int t = threadIdx.x; t < words; t += 256) stage[t] = src[t];
    }
    __syncthreads();

    const int64_t i = i0 + threadIdx.x;
    const bool live = i < n;
    const int64_t b = live ? off[i] : 0, e = live ? off[i + 1] : 0;
    const int R = P.hi - P.lo + 1;
    uint64_t *out = slots + b * R;
    const int w = P.w;
    constexpr int kBits = (int)sizeof(CODE) * 8;
    const CODE full_mask = (CODE)((CODE)~(CODE)0 >> (kBits - P.hi * w));   // low hi*w bits (1 <= hi*w <= kBits)
    CODE win = 0;
    int run = 0, cnt = 0;
    bool pending_space = false, any = false;

    auto feed = [&](uint32_t m, bool breaks) {
        win = (CODE)(((win << w) | (CODE)m) & full_mask);
        run = breaks ? 0 : (run < P.hi ? run + 1 : P.hi);
        for (int nn = P.lo; nn <= P.hi; ++nn) {
            if (run >= nn) {
                const CODE code = (CODE)((win & (CODE)((CODE)~(CODE)0 >> (kBits - nn * w))) << ((P.hi - nn) * w));
                out[cnt++] = code;
                if (bitmap) {
                    // test before set: almost every n-gram occurrence finds its bit already there, and the
                    // hot words ('inc', 'llc', ...) would otherwise serialise thousands of atomics.  The probe
                    // is a RELAXED ATOMIC load: a `volatile` one made the compiler drain vmcnt -- i.e. wait for
                    // the global store of the code just emitted -- before every probe (0.6 us per n-gram).
                    const uint32_t bit = 1u << ((uint32_t)code & 31u);
                    if (LB) {
                        uint32_t *wp = &lbm[code >> 5];
                        if (!(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit)) atomicOr(wp, bit);
                    } else {
                        uint32_t *wp = &bitmap[code >> 5];
                        if (!(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(wp, bit);
                    }
                }
            }
        }
    };

    // The walk is instantiated once per address space of the characters.  (A fetch helper with an
    // `if (staged)` inside was compiled to ONE flat load, and a flat load waits for vmcnt as well as
    // lgkmcnt: every character waited for the global stores of the codes emitted before it.)
    auto walk = [&](auto staged_tag) {
    constexpr bool kStaged = decltype(staged_tag)::value;
    for (int64_t p = b; p < e; ++p) {
        uint32_t c;
        if constexpr (kStaged) {
            const int o = (int)(p * CW - base4);
            c = CW == 1 ? (uint32_t)((const uint8_t *)stage)[o] : stage[o >> 2];
        } else {
            c = CW == 1 ? (uint32_t)((const uint8_t *)chars_v)[p] : ((const uint32_t *)chars_v)[p];
        }
        if (P.clean) {
            // reference _tfidf.py:142-146 on code points <= 0xFF: lower(), keep
            // [a-z0-9 ], collapse runs of ' ', strip
            if (c >= 'A' && c <= 'Z') c += 32;
            if (c == ' ') {
                if (any) pending_space = true;
                continue;
            }
            const bool letter = c >= 'a' && c <= 'z';
            const bool digit = c >= '0' && c <= '9';
            if (!letter && !digit) continue;
            if (pending_space) {
                feed(1u, P.remove_space != 0);
                pending_space = false;
            }
            any = true;
            feed(letter ? c - 'a' + 12u : c - '0' + 2u, false);
        } else {
            const uint32_t m = (int64_t)c < P.alpha_len ? alpha_map[c] : 0u;
            feed(m, m == 0u || (P.remove_space && c == ' '));
        }
    }
    };
    if (staged) walk(std::true_type{});
    else walk(std::false_type{});
    if (live) row_cnt[i] = cnt;
    if (LB && bitmap) {
        __syncthreads();
        // the global words are fetched unconditionally (coalesced, independent loads): a load behind
        // `if (wv)` made every non-zero word a dependent L2 round trip
#pragma unroll 8
        for (int t = threadIdx.x; t < P.bitmap_words; t += 256) {
            const uint32_t wv = lbm[t], gv = bitmap[t];
            if (wv & ~gv) atomicOr(&bitmap[t], wv);
        }
    }
}

// Document-frequency counters.  A hot n-gram ('inc': 20 % of all names) would
// serialise tens of thousands of atomics on ONE address, so every n-gram has
// 2^shift counters selected by a salt (the row); k_df_reduce adds them up.
struct DfSink {
    int32_t *p;   // [vocab << shift], or NULL: do not count
    int shift;
    __device__ inline void add(uint32_t key, int64_t salt) const
    {
        if (p) atomicAdd(&p[((int64_t)key << shift) + (salt & ((1 << shift) - 1))], 1);
    }
};

// One string of 1 .. kShortMax (128) n-gram codes in its slots -> the (column id, count) pairs of its distinct known n-grams,
// ascending, in the same slots; one WAVE, no barrier (k_rows_short runs it on a wave per string; k_extract_wave<.., ROWS>
// right behind the extraction of the string, by the wave that extracted it).
__device__ inline void rows_short_row(const VocabView &V, uint64_t *base, int cnt, int lane, int32_t *row_nnz_at, int64_t row, DfSink df)
{
    if (cnt > 64) {
        // 65 .. 128 n-grams: two keys per lane (elements lane and lane + 64 of a 128-key bitonic network), still one wave and
        // no barrier.  (k_rows_long sorts in LDS with a workgroup barrier per stage: ~30 us for ONE such string, and a list of
        // names has a few -- 21 of the 100 000 company names, none beyond 87 n-grams -- so that launch was pure latency.)
        uint32_t k0 = vocab_rank(V, base[lane]);
        uint32_t k1 = lane + 64 < cnt ? vocab_rank(V, base[lane + 64]) : kInvalid;
#pragma unroll
        for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j == 64) {                      // (k == 128) the partner is the lane's other key; the last merge ascends
                    const uint32_t mn = k0 < k1 ? k0 : k1, mx = k0 < k1 ? k1 : k0;
                    k0 = mn;
                    k1 = mx;
                } else {
                    const uint32_t o0 = __shfl_xor(k0, j, 64), o1 = __shfl_xor(k1, j, 64);
                    const bool lower = (lane & j) == 0;
                    const bool asc0 = (lane & k) == 0, asc1 = ((lane + 64) & k) == 0;     // element index & k
                    const uint32_t mn0 = k0 < o0 ? k0 : o0, mx0 = k0 < o0 ? o0 : k0;
                    const uint32_t mn1 = k1 < o1 ? k1 : o1, mx1 = k1 < o1 ? o1 : k1;
                    k0 = (lower == asc0) ? mn0 : mx0;
                    k1 = (lower == asc1) ? mn1 : mx1;
                }
            }
        }
        // run lengths over the 128 sorted keys (element e = lane + 64 r)
        const uint32_t p0 = __shfl_up(k0, 1, 64), last0 = __shfl(k0, 63, 64);
        uint32_t p1 = __shfl_up(k1, 1, 64);
        if (lane == 0) p1 = last0;
        const bool v0 = k0 != kInvalid, v1 = k1 != kInvalid;
        const bool h0 = v0 && (lane == 0 || k0 != p0), h1 = v1 && k1 != p1;
        const uint64_t H0 = __ballot(h0), H1 = __ballot(h1);
        const int nvalid = __popcll(__ballot(v0)) + __popcll(__ballot(v1));
        const uint64_t below = (1ull << lane) - 1ull;
        const uint64_t above0 = lane == 63 ? 0ull : (H0 >> (lane + 1)), above1 = lane == 63 ? 0ull : (H1 >> (lane + 1));
        if (h0) {
            const int next = above0 ? lane + 1 + __builtin_ctzll(above0) : (H1 ? 64 + __builtin_ctzll(H1) : nvalid);
            base[__popcll(H0 & below)] = ((uint64_t)(uint32_t)(next - lane) << 32) | k0;
            df.add(k0, row);
        }
        if (h1) {
            const int e = lane + 64;
            const int next = above1 ? e + 1 + __builtin_ctzll(above1) : nvalid;
            base[__popcll(H0) + __popcll(H1 & below)] = ((uint64_t)(uint32_t)(next - e) << 32) | k1;
            df.add(k1, row);
        }
        if (lane == 0) *row_nnz_at = __popcll(H0) + __popcll(H1);
        return;
    }
    uint32_t key = kInvalid;
    if (lane < cnt) key = vocab_rank(V, base[lane]);
    // 64-lane bitonic sort, ascending
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint32_t o = __shfl_xor(key, j, 64);
            const bool asc = (lane & k) == 0;
            const bool lower = (lane & j) == 0;
            const uint32_t mn = key < o ? key : o, mx = key < o ? o : key;
            key = (lower == asc) ? mn : mx;
        }
    }
    const uint32_t prev = __shfl_up(key, 1, 64);
    const bool valid = key != kInvalid;
    const bool head = valid && (lane == 0 || key != prev);
    const uint64_t H = __ballot(head);
    const int nvalid = __popcll(__ballot(valid));
    if (head) {
        const int pos = __popcll(H & ((1ull << lane) - 1ull));
        const uint64_t above = lane == 63 ? 0ull : (H >> (lane + 1));
        const int next = above ? lane + 1 + __builtin_ctzll(above) : nvalid;
        base[pos] = ((uint64_t)(uint32_t)(next - lane) << 32) | key;   // (.x = id, .y = tf) little-endian
        df.add(key, row);
    }
    if (lane == 0) *row_nnz_at = __popcll(H);
}

// ---------------------------------------------------------------------------
// k_extract_wave: one WAVE per string, one LANE per character (round 4).
// ---------------------------------------------------------------------------
// The thread-per-string kernel above runs as long as the longest string of a wave takes (~0.25 us per character of
// divergent clean / window code): 60 us for a list of 10 000 names that keeps 40 CUs busy.  Here a wave takes a string
// 64 characters at a time: every lane classifies its character, ballots + bit counts give each kept character its
// place in the cleaned sequence (a ' ' is put in front of it when spaces lie between it and the kept character before),
// the cleaned symbols go to an LDS line, and then every lane builds the n-grams that END at its position -- the last
// `hi` symbols, the distance to the last breaking symbol -- and the codes are compacted into the string's slots by ballot.
// Same codes, same counts as k_extract (the order inside a string's slots differs: k_rows_* sort them anyway).
// For lists of up to 32 768 strings whose longest has at most kWaveMaxLen characters; the others take k_extract.
constexpr int kWaveMaxLen = 256;
constexpr int kWaveStringsMax = 128;       // strings per workgroup at most: one LDS bitmap, one staged stretch of characters

// ROWS (a transform: the vocabulary is there): the wave goes on to sort, look up and count the string's n-grams itself
// (rows_short_row; strings of more than kShortMax n-grams are left to k_rows_long) -- a launch less in a chain whose cost on
// short lists IS the number of its launches.
template <int CW, bool LB, typename CODE, bool ROWS = false>
__global__ __launch_bounds__(256) void k_extract_wave(const void *__restrict__ chars_v, const int64_t *__restrict__ off,
                                                       int64_t n, ExtractParams P, const uint32_t *__restrict__ alpha_map,
                                                       uint64_t *__restrict__ slots, int32_t *__restrict__ row_cnt,
                                                       uint32_t *__restrict__ bitmap, int32_t per_wg, VocabView V = VocabView{},
                                                       int32_t *__restrict__ row_nnz = nullptr, ListB B = ListB{})
{
    unsigned bx = blockIdx.x;
    if (bx >= B.grid0) {           // (a workgroup of the second list)
        bx -= B.grid0;
        chars_v = B.chars;
        off = B.off;
        n = B.n;
        slots = B.slots;
        row_cnt = B.row_cnt;
    }
    // (8 KB of staged characters: 32 strings of 256.  It was 24 KB -- with the 32 KB bitmap copy of a fit 60 KB per workgroup, two
    // per CU: the 626 workgroups of config 2's two lists in one launch took two rounds, 66 us; at 44 KB three fit and they take one)
    constexpr int kStageBytes = 8 * 1024;
    __shared__ uint32_t stage[kStageBytes / 4];
    __shared__ uint32_t lbm[LB ? kLdsBitmapWords : 1];
    __shared__ uint32_t s_sym[4][kWaveMaxLen];          // cleaned symbols; bit 31 = breaks every window it is in
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (LB)
        for (int t = threadIdx.x; t < P.bitmap_words; t += 256) lbm[t] = 0u;
    const int64_t i0 = (int64_t)bx * per_wg;
    const int64_t i_end = i0 + per_wg < n ? i0 + per_wg : n;
    const int64_t byte0 = off[i0] * CW, byte1 = off[i_end] * CW;
    const int64_t base4 = byte0 & ~(int64_t)3;
    const bool staged = byte1 - base4 <= kStageBytes;
    if (staged) {
        const uint32_t *src = (const uint32_t *)((const uint8_t *)chars_v + base4);
        const int words = (int)((byte1 - base4 + 3) >> 2);     // the buffer is padded by 16 bytes
        for (int t = threadIdx.x; t < words; t += 256) stage[t] = src[t];
    }
    __syncthreads();
    const int R = P.hi - P.lo + 1, w = P.w, hi = P.hi;
    constexpr int kBits = (int)sizeof(CODE) * 8;
    uint32_t *sym = s_sym[wave];
    const uint64_t below = (1ull << lane) - 1ull;       // lanes before this one

    for (int64_t i = i0 + wave; i < i_end; i += 4) {
        const int64_t b = off[i], e = off[i + 1];
        const int len = (int)(e - b);
        // ---- pass 1: the cleaned symbol sequence -------------------------------------------------------------------
        int n_out = 0;
        bool pend = false, any = false;                 // a ' ' since the last kept character / any kept character so far
        for (int c0 = 0; c0 < len; c0 += 64) {
            const int p = c0 + lane;
            uint32_t c = 0xffffffffu;
            if (p < len) {
                if (staged) {
                    const int o = (int)((b + p) * CW - base4);
                    c = CW == 1 ? (uint32_t)((const uint8_t *)stage)[o] : stage[o >> 2];
                } else {
                    c = CW == 1 ? (uint32_t)((const uint8_t *)chars_v)[b + p] : ((const uint32_t *)chars_v)[b + p];
                }
            }
            if (P.clean) {
                // reference _tfidf.py:142-146 on code points <= 0xFF: lower(), keep [a-z0-9 ], collapse runs of ' ', strip
                if (c >= 'A' && c <= 'Z') c += 32;
                const bool letter = c >= 'a' && c <= 'z', digit = c >= '0' && c <= '9';
                const bool keep = letter || digit;
                const uint64_t keepm = __ballot(keep), spm = __ballot(c == ' ');
                const uint64_t kb = keepm & below;                                     // kept characters before this one
                const int q = kb ? 63 - __clzll((long long)kb) : -1;                   // ... the nearest of them
                const uint64_t between = spm & below & ~((q >= 0 ? (2ull << q) : 1ull) - 1ull);      // spaces in (q, lane)
                const bool first = kb == 0ull && !any;                                 // the string's first kept character
                const bool sp_before = keep && !first && (between != 0ull || (kb == 0ull && pend));
                const uint64_t esm = __ballot(sp_before);
                const int pos = n_out + __popcll(kb) + __popcll(esm & (below | (1ull << lane)));
                if (sp_before) sym[pos - 1] = 1u | (P.remove_space ? 0x80000000u : 0u);
                if (keep) sym[pos] = letter ? c - 'a' + 12u : c - '0' + 2u;
                n_out += __popcll(keepm) + __popcll(esm);
                if (keepm) {
                    const int last = 63 - __clzll((long long)keepm);
                    pend = last < 63 && (spm >> (last + 1)) != 0ull;
                    any = true;
                } else {
                    pend = pend || spm != 0ull;
                }
            } else {
                // every character is a symbol: its rank in the fitted alphabet (0 = unknown: breaks), ' ' breaks when asked to
                if (p < len) {
                    const uint32_t m = (int64_t)c < P.alpha_len ? alpha_map[c] : 0u;
                    sym[p] = m | ((m == 0u || (P.remove_space && c == ' ')) ? 0x80000000u : 0u);
                }
                n_out = len;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- pass 2: the n-grams that end at every position ---------------------------------------------------------
        uint64_t *out = slots + b * R;
        int cnt = 0;
        for (int p0 = 0; p0 < n_out; p0 += 64) {
            const int p = p0 + lane;
            CODE win = 0;
            int run = 0;                                   // symbols ending at p without a break (at most hi)
            bool open = p < n_out;
            for (int j = 0; j < hi; ++j) {
                uint32_t sj = 0x80000000u;
                if (p - j >= 0 && p < n_out) sj = sym[p - j];
                open = open && !(sj & 0x80000000u);
                run += open ? 1 : 0;
                win |= (CODE)((CODE)(sj & 0x7fffffffu) << (j * w));
            }
            for (int nn = P.lo; nn <= hi; ++nn) {
                const bool pred = run >= nn;
                const uint64_t mk = __ballot(pred);
                if (pred) {
                    const CODE code = (CODE)((win & (CODE)((CODE)~(CODE)0 >> (kBits - nn * w))) << ((hi - nn) * w));
                    out[cnt + __popcll(mk & below)] = code;
                    if (bitmap) {
                        const uint32_t bit = 1u << ((uint32_t)code & 31u);
                        if (LB) {
                            uint32_t *wp = &lbm[code >> 5];
                            if (!(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit)) atomicOr(wp, bit);
                        } else {
                            uint32_t *wp = &bitmap[code >> 5];
                            if (!(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(wp, bit);
                        }
                    }
                }
                cnt += __popcll(mk);
            }
        }
        if (lane == 0) row_cnt[i] = cnt;
        if (ROWS) {
            // (the codes the wave's lanes just stored are read back by other lanes of the same wave)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (cnt == 0) {
                if (lane == 0) row_nnz[i] = 0;
            } else if (cnt <= kShortMax) {
                rows_short_row(V, out, cnt, lane, row_nnz + i, i, DfSink{nullptr, 0});
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();               // the symbol line is reused by the wave's next string
    }
    if (LB && bitmap) {
        __syncthreads();
#pragma unroll 8
        for (int t = threadIdx.x; t < P.bitmap_words; t += 256) {
            const uint32_t wv = lbm[t], gv = bitmap[t];
            if (wv & ~gv) atomicOr(&bitmap[t], wv);
        }
    }
}

// Distinct code points of a list -> bitmap over the Unicode range.  Code points below 2048 (every Latin
// text) are collected in LDS first and flushed once per workgroup; the others probe the global word before
// setting it.  (One unconditional global atomicOr per character meant millions of atomics on the three or
// four words that hold the ASCII letters: 47 ms per fit at 100k + 100k names.)
template <int CW>
__global__ __launch_bounds__(256) void k_alpha_mark(const void *__restrict__ chars_v, int64_t n_units,
                                                     uint32_t *__restrict__ present)
{
    constexpr int kLowWords = 2048 / 32;
    __shared__ uint32_t low[kLowWords];
    if (threadIdx.x < kLowWords) low[threadIdx.x] = 0u;
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_units; p += (int64_t)gridDim.x * 256) {
        uint32_t c = CW == 1 ? (uint32_t)((const uint8_t *)chars_v)[p] : ((const uint32_t *)chars_v)[p];
        const uint32_t bit = 1u << (c & 31u);
        if (c < 2048u) {
            if (!(__hip_atomic_load(&low[c >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit)) atomicOr(&low[c >> 5], bit);
        } else if (c < 0x110000u) {
            if (!(__hip_atomic_load(&present[c >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(&present[c >> 5], bit);
        }
    }
    __syncthreads();
    if (threadIdx.x < kLowWords) {
        const uint32_t w = low[threadIdx.x];
        if (w && (w & ~__hip_atomic_load(&present[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
            atomicOr(&present[threadIdx.x], w);
    }
}

__global__ __launch_bounds__(256) void k_group_popc(const uint32_t *__restrict__ bitmap, int64_t n_groups,
                                                     int32_t *__restrict__ prefix)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups) return;
    const uint4 *w4 = (const uint4 *)(bitmap + g * 8);
    const uint4 a = w4[0], b = w4[1];
    prefix[g] = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) +
                __popc(b.w);
}

// ... and, for bitmaps of up to 1024 groups (cleaned 3-grams: 2^18 codes, exactly 1024), the exclusive scan of the counts in the
// same launch -- one workgroup, a group per thread: prefix[g] = n-grams before group g, prefix[n_groups] = the vocabulary's size,
// which also goes to the host's pinned word (LazyI32) as the scan kernel's total does
__global__ __launch_bounds__(1024) void k_group_popc_scan(const uint32_t *__restrict__ bitmap, int32_t n_groups,
                                                           int32_t *__restrict__ prefix, int32_t *__restrict__ host_total)
{
    __shared__ int32_t wsum[16];
    const int g = threadIdx.x, lane = g & 63, wave = g >> 6;
    int c = 0;
    if (g < n_groups) {
        const uint4 *w4 = (const uint4 *)(bitmap + (int64_t)g * 8);
        const uint4 a = w4[0], b = w4[1];
        c = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
    }
    int inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (g < n_groups) prefix[g] = base + inc - c;
    if (g == n_groups - 1) {
        prefix[n_groups] = base + inc;
        if (host_total) __hip_atomic_store(host_total, base + inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void k_set_bits(const uint64_t *__restrict__ codes, int64_t n,
                                                   uint32_t *__restrict__ bitmap)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t code = codes[i];
    atomicOr(&bitmap[code >> 5], 1u << ((uint32_t)code & 31u));
}

// ---- sorted-vocabulary mode: gather every emitted code, sort, keep the distinct ones ----------------
// 16 lanes per string: copy the string's row_cnt codes from its slot range to dst[code_off[i] ...]
__global__ __launch_bounds__(256) void k_gather_codes(const int64_t *__restrict__ off, int64_t n, int32_t R,
                                                       const uint64_t *__restrict__ slots,
                                                       const int32_t *__restrict__ code_off,
                                                       uint64_t *__restrict__ dst)
{
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    if (row >= n) return;
    const int o = code_off[row], cnt = code_off[row + 1] - o;
    const uint64_t *src = slots + off[row] * R;
    for (int t = sub; t < cnt; t += 16) dst[o + t] = src[t];
}

__global__ __launch_bounds__(256) void k_flag_heads(const uint64_t *__restrict__ sorted, int64_t n,
                                                     int32_t *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = i == 0 || sorted[i] != sorted[i - 1];
}

__global__ __launch_bounds__(256) void k_scatter_heads(const uint64_t *__restrict__ sorted, int64_t n,
                                                        const int32_t *__restrict__ pos /* exclusive scan of flags */,
                                                        uint64_t *__restrict__ vcodes)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && (i == 0 || sorted[i] != sorted[i - 1])) vcodes[pos[i]] = sorted[i];
}

// codes of the vocabulary in column order
__global__ __launch_bounds__(256) void k_export_codes(const uint32_t *__restrict__ bitmap,
                                                       const int32_t *__restrict__ prefix, int64_t n_groups,
                                                       uint64_t *__restrict__ codes)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups) return;
    if (prefix[g + 1] == prefix[g]) return;
    int64_t o = prefix[g];
    for (int j = 0; j < 8; ++j) {
        uint32_t word = bitmap[g * 8 + j];
        while (word) {
            const int bit = __builtin_ctz(word);
            word &= word - 1;
            codes[o++] = ((uint64_t)g << 8) | (uint64_t)(j * 32 + bit);
        }
    }
}

// ---------------------------------------------------------------------------
// k_rows_short: one wave per string with <= 64 n-grams.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_df_reduce(const int32_t *__restrict__ sharded, int64_t vocab, int shift,
                                                    int32_t *__restrict__ df)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= vocab) return;
    int32_t s = 0;
    for (int j = 0; j < (1 << shift); ++j) s += sharded[(k << shift) + j];
    df[k] = s;
}

// Document frequencies WITHOUT global atomics, for vocabularies of up to 2 * kHistWords n-grams: one
// workgroup per kHistRows strings counts the strings' distinct column ids (the (id, tf) pairs
// k_rows_* left in the slots) in an LDS histogram, two 16-bit counters per word, and writes the
// histogram out as one dense row of `partial`; k_df_hist_reduce adds the rows up.  (1.4 M device-scope
// atomics, even spread over 32 counters per n-gram, were 70 % of the vectoriser's run time.)
constexpr int kHistRows = 1024;   // <= 65 535, the range of a 16-bit counter; small enough for >= 1 workgroup per CU at 100k strings

__global__ __launch_bounds__(1024) void k_df_hist(const int64_t *__restrict__ off, int64_t n, int32_t R,
                                                   const uint64_t *__restrict__ slots,
                                                   const int32_t *__restrict__ row_nnz, int32_t words,
                                                   uint32_t *__restrict__ partial, ListB B = ListB{})
{
    __shared__ uint32_t h[kHistWords];
    for (int t = threadIdx.x; t < words; t += 1024) h[t] = 0u;
    __syncthreads();
    unsigned bx = blockIdx.x;
    if (bx >= B.grid0) {           // (the second list's chunks follow the first's in `partial`: blockIdx.x numbers them all)
        bx -= B.grid0;
        off = B.off;
        n = B.n;
        slots = B.slots;
        row_nnz = B.row_cnt + (B.n + 1);
    }
    const int64_t r0 = (int64_t)bx * kHistRows;
    const int64_t r1 = r0 + kHistRows < n ? r0 + kHistRows : n;
    // a 16-lane group takes 16 consecutive strings at a time: the lanes fetch the 16 strings' lengths
    // and slot addresses in parallel, then the first 16 entries of all 16 strings are loaded back to back
    // (a group walking its strings one by one paid three dependent L2 round trips per string)
    const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int lane0 = (threadIdx.x & 63) & ~15;
    for (int64_t rb = r0 + grp * 16; rb < r1; rb += 64 * 16) {   // (uniform trip count per wave: r1 - r0 <= 1024)
        const int64_t mine = rb + sub;
        int nn = 0;
        int64_t so = 0;
        if (mine < r1) {
            nn = row_nnz[mine];
            so = off[mine] * R;
        }
        uint32_t kk[16];
        int nni[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            nni[i] = __shfl(nn, lane0 + i, 64);
            const int64_t soi = ((int64_t)__shfl((int)(so >> 32), lane0 + i, 64) << 32) |
                                (uint32_t)__shfl((int)(uint32_t)so, lane0 + i, 64);
            kk[i] = sub < nni[i] ? ((const uint2 *)(slots + soi))[sub].x : 0u;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (sub < nni[i]) atomicAdd(&h[kk[i] >> 1], 1u << ((kk[i] & 1u) * 16));
        for (int i = 0; i < 16; ++i) {                            // strings with more than 16 distinct n-grams
            const int n_i = __shfl(nn, lane0 + i, 64);
            if (n_i <= 16) continue;
            const int64_t soi = ((int64_t)__shfl((int)(so >> 32), lane0 + i, 64) << 32) |
                                (uint32_t)__shfl((int)(uint32_t)so, lane0 + i, 64);
            const uint2 *in = (const uint2 *)(slots + soi);
            for (int t = sub + 16; t < n_i; t += 16) {
                const uint32_t k = in[t].x;
                atomicAdd(&h[k >> 1], 1u << ((k & 1u) * 16));
            }
        }
    }
    __syncthreads();
    uint32_t *dst = partial + (int64_t)blockIdx.x * words;
    for (int t = threadIdx.x; t < words; t += 1024) dst[t] = h[t];
}

// (idf != NULL -- one GPU: the number of documents is known -- the idf of the two n-grams is written here too: k_idf's
// formula, one launch less on a path whose short lists pay per launch)
__global__ __launch_bounds__(256) void k_df_hist_reduce(const uint32_t *__restrict__ partial, int32_t words,
                                                         int32_t chunks, int64_t vocab, int32_t *__restrict__ df,
                                                         double n_docs = 0.0, double *__restrict__ idf = nullptr)
{
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= words) return;
    int32_t lo = 0, hi = 0;
#pragma unroll 8
    for (int c = 0; c < chunks; ++c) {   // (unrolled: eight independent loads in flight)
        const uint32_t v = partial[(int64_t)c * words + w];
        lo += (int32_t)(v & 0xffffu);
        hi += (int32_t)(v >> 16);
    }
    df[2 * (int64_t)w] = lo;
    if (2 * (int64_t)w + 1 < vocab) df[2 * (int64_t)w + 1] = hi;
    if (idf) {
        idf[2 * (int64_t)w] = log((n_docs + 1.0) / ((double)lo + 1.0)) + 1.0;
        if (2 * (int64_t)w + 1 < vocab) idf[2 * (int64_t)w + 1] = log((n_docs + 1.0) / ((double)hi + 1.0)) + 1.0;
    }
}

__global__ __launch_bounds__(256) void k_rows_short(const int64_t *__restrict__ off, int64_t n, int32_t R,
                                                     VocabView V, uint64_t *__restrict__ slots,
                                                     const int32_t *__restrict__ row_cnt,
                                                     int32_t *__restrict__ row_nnz, DfSink df, ListB B = ListB{})
{
    const int lane = threadIdx.x & 63;
    unsigned bx = blockIdx.x;
    if (bx >= B.grid0) {
        bx -= B.grid0;
        off = B.off;
        n = B.n;
        slots = B.slots;
        row_cnt = B.row_cnt;
        row_nnz = B.row_cnt + (B.n + 1);
    }
    const int64_t row = (int64_t)bx * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int cnt = row_cnt[row];
    if (cnt > kShortMax) return;  // k_rows_long
    if (cnt == 0) {
        if (lane == 0) row_nnz[row] = 0;
        return;
    }
    rows_short_row(V, slots + off[row] * R, cnt, lane, row_nnz + row, row, df);
}

// ---------------------------------------------------------------------------
// k_rows_long: one workgroup per string with > kShortMax (128) n-grams.
//   cnt <= kLongMax : keys sorted in LDS; otherwise in a global scratch slab.
// ---------------------------------------------------------------------------
// Sort + run-length encode one long string.  KEYS is either an LDS array or a
// global scratch slab; the body is instantiated once per address space (a
// runtime-selected flat pointer is avoided on purpose).
template <typename KEYS>
__device__ inline void rows_long_body(KEYS keys, int cnt, const VocabView &V, uint64_t *base, int32_t *row_nnz_out,
                                      DfSink df, int *sh_heads, int *sh_valid)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int npow2 = 128;
    while (npow2 < cnt) npow2 <<= 1;
    for (int t = threadIdx.x; t < npow2; t += 256) keys[t] = t < cnt ? vocab_rank(V, base[t]) : kInvalid;
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < npow2; t += 256) {
                const int x = t ^ j;
                if (x > t) {
                    const bool asc = (t & k) == 0;
                    const uint32_t a = keys[t], b = keys[x];
                    if ((a > b) == asc) {
                        keys[t] = b;
                        keys[x] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    // run-length encode: wave 0 walks the sorted keys, writes (id, start)
    uint2 *out = (uint2 *)base;
    if (wave == 0) {
        int nheads = 0, nvalid = 0;
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int t = c0 + lane;
            uint32_t key = kInvalid, prev = kInvalid;
            if (t < cnt) key = keys[t];
            if (t > 0 && t < cnt) prev = keys[t - 1];
            const bool valid = key != kInvalid;
            const bool head = valid && (t == 0 || key != prev);
            const uint64_t H = __ballot(head);
            if (head) {
                out[nheads + __popcll(H & ((1ull << lane) - 1ull))] = make_uint2(key, (uint32_t)t);
                df.add(key, blockIdx.x);
            }
            nheads += __popcll(H);
            nvalid += __popcll(__ballot(valid));
        }
        if (lane == 0) {
            *sh_heads = nheads;
            *sh_valid = nvalid;
            *row_nnz_out = nheads;
        }
    }
    __syncthreads();
    const int nheads = *sh_heads, nvalid = *sh_valid;
    for (int c0 = 0; c0 < nheads; c0 += 256) {   // start -> tf, chunk by chunk (read, barrier, write)
        const int p = c0 + threadIdx.x;
        uint32_t tf = 0;
        if (p < nheads) tf = (p + 1 < nheads ? out[p + 1].y : (uint32_t)nvalid) - out[p].y;
        __syncthreads();
        if (p < nheads) out[p].y = tf;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_rows_long(const int64_t *__restrict__ off, int64_t n, int32_t R, VocabView V,
                                                    uint64_t *__restrict__ slots,
                                                    const int32_t *__restrict__ row_cnt,
                                                    int32_t *__restrict__ row_nnz, DfSink df,
                                                    uint32_t *__restrict__ giant_scratch, int64_t giant_stride)
{
    __shared__ uint32_t lds_keys[kLongMax];
    __shared__ int sh_heads, sh_valid, n_long;
    __shared__ int long_rows[kLongTile];
    // tiles of kLongTile strings: the threads look at one string each (a workgroup walking the strings one
    // by one spent its time on dependent loads that almost always said "short"), then the workgroup handles
    // the tile's long strings one after the other.  Small tiles: a list of long documents must still
    // spread over all workgroups.
    const int64_t n_tiles = (n + kLongTile - 1) / kLongTile;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (threadIdx.x == 0) n_long = 0;
        __syncthreads();
        const int64_t mine = tile * kLongTile + threadIdx.x;
        if ((int)threadIdx.x < kLongTile && mine < n && row_cnt[mine] > kShortMax) long_rows[atomicAdd(&n_long, 1)] = (int)threadIdx.x;
        __syncthreads();
        const int m = n_long;
        for (int i = 0; i < m; ++i) {
            const int64_t row = tile * kLongTile + long_rows[i];
            const int cnt = row_cnt[row];
            uint64_t *base = slots + off[row] * R;
            if (cnt <= kLongMax)
                rows_long_body(&lds_keys[0], cnt, V, base, row_nnz + row, df, &sh_heads, &sh_valid);
            else
                rows_long_body(giant_scratch + (int64_t)blockIdx.x * giant_stride, cnt, V, base, row_nnz + row, df,
                               &sh_heads, &sh_valid);
            __syncthreads();
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_idf(const int32_t *__restrict__ df, int64_t vocab, double n_docs,
                                              double *__restrict__ idf)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= vocab) return;
    // sklearn text.py:1662-1679 (smooth_idf): ln((1+n)/(1+df)) + 1
    idf[k] = log((n_docs + 1.0) / ((double)df[k] + 1.0)) + 1.0;
}

// 16 lanes per string: tf*idf in float64, the sum of squares accumulated SEQUENTIALLY in index order
// (every lane replays the same additions, so the result is bit-identical to sklearn's
// inplace_csr_row_normalize_l2 loop), sqrt, divide, round to fp32.  The lanes fetch their (id, tf)
// pair and idf in parallel; a thread per string walked them as ~14 dependent L2 round trips.
__device__ inline double shfl_f64(double v, int src_lane)
{
    const int lo = __shfl(__double2loint(v), src_lane, 64), hi = __shfl(__double2hiint(v), src_lane, 64);
    return __hiloint2double(hi, lo);
}

// SELF_SCAN (lists of up to kSelfScanRows strings): the row offsets are not there yet -- every workgroup adds up the counts of
// the rows before its own (a few KB from L2), writes the offsets of its 16 rows, the last one the total (to indptr[n] and to
// the host's pinned word, LazyI32) -- one launch where a copy, a scan and this kernel were three: a transform of a short list
// is a chain of short dependent launches, and what it costs is their number (5 - 7 us each).
constexpr int64_t kSelfScanRows = 16384;
template <bool SELF_SCAN>
__global__ __launch_bounds__(256) void k_finalize(const int64_t *__restrict__ off, int64_t n, int32_t R,
                                                   const uint64_t *__restrict__ slots,
                                                   int32_t *__restrict__ indptr, const double *__restrict__ idf,
                                                   int32_t *__restrict__ indices, float *__restrict__ data,
                                                   const int32_t *__restrict__ row_nnz, int32_t *__restrict__ host_total)
{
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    const int lane0 = (threadIdx.x & 63) & ~15;    // first lane of this row's group inside the wave
    const bool live = row < n;
    const uint2 *in = live ? (const uint2 *)(slots + off[row] * R) : nullptr;
    int o, nn;
    if (SELF_SCAN) {
        __shared__ int32_t s_part[4], s_cnt[16];
        const int64_t row0 = (int64_t)blockIdx.x * 16;
        int32_t acc = 0;
        for (int64_t t = threadIdx.x; t < row0; t += 256) acc += row_nnz[t];
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
        if (threadIdx.x < 16) s_cnt[threadIdx.x] = row0 + threadIdx.x < n ? row_nnz[row0 + threadIdx.x] : 0;
        __syncthreads();
        o = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        const int mine = (int)(row - row0);
        for (int k = 0; k < mine; ++k) o += s_cnt[k];
        nn = live ? s_cnt[mine] : 0;
        if (live && sub == 0) indptr[row] = o;
        if (row == n - 1 && sub == 0) {
            indptr[n] = o + nn;
            if (host_total) __hip_atomic_store(host_total, o + nn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else {
        o = live ? indptr[row] : 0;
        nn = live ? indptr[row + 1] - o : 0;
    }
    // rows of one wave may differ in length: every lane runs the longest row's trip count so that the
    // shuffles stay convergent
    int nn_max = nn;
    for (int d = 16; d < 64; d <<= 1) nn_max = max(nn_max, __shfl_xor(nn_max, d, 64));
    double ss = 0.0;
    for (int c0 = 0; c0 < nn_max; c0 += 16) {          // pass 1: ss
        const int t = c0 + sub;
        double v = 0.0;
        if (t < nn) {
            const uint2 e = in[t];
            v = __dmul_rn((double)e.y, idf[e.x]);
        }
        const double v2 = __dmul_rn(v, v);
        const int m = min(16, nn_max - c0);
        for (int j = 0; j < m; ++j) {
            const double term = shfl_f64(v2, lane0 + j);
            if (c0 + j < nn) ss = __dadd_rn(ss, term);
        }
    }
    const double nrm = sqrt(ss);
    for (int t = sub; t < nn; t += 16) {                // pass 2: normalise and store
        const uint2 e = in[t];
        double v = __dmul_rn((double)e.y, idf[e.x]);
        if (ss != 0.0) v = v / nrm;
        indices[o + t] = (int32_t)e.x;
        data[o + t] = (float)v;
    }
}

__global__ __launch_bounds__(256) void k_copy_i32(const int32_t *__restrict__ src, int64_t n, int32_t *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ---------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------
static inline unsigned grid_for(int64_t n, int per_block = 256) { return (unsigned)((n + per_block - 1) / per_block); }

static int bits_for(uint64_t max_value)
{
    int b = 1;
    while ((max_value >> b) != 0) ++b;
    return b;
}

static int prepare_slots(pfz_ctx *ctx, const pfz_tfidf *v, pfz_strings *s)
{
    const int R = v->params.ngram_hi - v->params.ngram_lo + 1;
    const size_t need = (size_t)(s->n_units > 0 ? s->n_units : 1) * (size_t)R;
    if (s->slots_cap < need) {
        if (s->slots) pool_free(s->slots);
        s->slots = nullptr;
        PFZ_TRY(pool_alloc(ctx, &s->slots, need * sizeof(uint64_t)));
        s->slots_cap = need;
    }
    if (!s->row_cnt) PFZ_TRY(pool_alloc(ctx, &s->row_cnt, (size_t)(s->n + 1) * 2 * sizeof(int32_t)));
    return PFZ_OK;
}

// sb (a fit over two lists): its strings are extracted by the same launch when both lists take the wave kernel and share a
// character width; *sb_done says whether that happened (else the caller extracts it on its own).
static int run_extract(pfz_ctx *ctx, const pfz_tfidf *v, pfz_strings *s, bool mark, bool *rows_done = nullptr,
                       pfz_strings *sb = nullptr, bool *sb_done = nullptr)
{
    if (sb_done) *sb_done = false;
    PFZ_TRY(prepare_slots(ctx, v, s));
    if (s->n == 0) return PFZ_OK;
    const int64_t bitmap_words = v->n_groups * 8;
    ExtractParams P{v->params.ngram_lo, v->params.ngram_hi, v->params.clean, v->params.remove_space_ngrams,
                    v->bits_per_char, (int32_t)(bitmap_words <= kLdsBitmapWords ? bitmap_words : 0), v->alpha_map_len};
    const bool lds_bitmap = mark && bitmap_words <= kLdsBitmapWords;
    ProfScope ps(ctx, "k1_extract");
    // a wave per string (a lane per character) when every string fits the wave's symbol line; PFZ_K1_EXTRACT=thread forces
    // the thread-per-string kernel (tests)
    const char *which = getenv("PFZ_K1_EXTRACT");
    const bool wave = s->max_len <= kWaveMaxLen && (which ? which[0] == 'w' : s->n <= 32768);
    // (10 000 + 10 000 names, both launches: thread per string 0.119 ms; a wave per string with 16 / 32 / 64 / 128 strings per
    // workgroup 0.091 / 0.068 / 0.077 / 0.104 ms.  At 100 000 strings the thread-per-string kernel wins, 0.071 against 0.115 ms:
    // a wave's strings are a chain of dependent LDS round trips, and there the chip is full of threads anyway)
    int per_wg = 32;
    if (const char *e = getenv("PFZ_K1_WAVE_STRINGS")) per_wg = std::max(4, std::min(kWaveStringsMax, atoi(e)));
    // the second list in the same launch (PFZ_K1_TWO_LISTS=0: never -- tests, A/B)
    ListB B;
    unsigned grid_b = 0;
    const char *two_knob = getenv("PFZ_K1_TWO_LISTS");
    if (sb && sb_done && wave && sb->n > 0 && sb->n <= 32768 && sb->max_len <= kWaveMaxLen && sb->char_width == s->char_width &&
        !(which && which[0] != 'w') && !(two_knob && atoi(two_knob) == 0)) {
        PFZ_TRY(prepare_slots(ctx, v, sb));
        B.chars = sb->chars;
        B.off = sb->offsets;
        B.n = sb->n;
        B.slots = sb->slots;
        B.row_cnt = sb->row_cnt;
        B.grid0 = grid_for(s->n, per_wg);
        grid_b = grid_for(sb->n, per_wg);
        *sb_done = true;
    }
    // a transform of a list the wave kernel takes: extraction and the short rows in ONE launch (see k_extract_wave<.., ROWS>)
    const char *fuse_knob = getenv("PFZ_K1_FUSE_ROWS");      // (tests: 0 = two launches)
    if (rows_done) *rows_done = false;
    if (rows_done && !mark && wave && !(fuse_knob && atoi(fuse_knob) == 0)) {
        VocabView V{v->bitmap, v->prefix, v->vcodes, v->vocab};
        int32_t *row_nnz = s->row_cnt + (s->n + 1);
#define PFZ_EXTRACT_ROWS(CW, CODE)                                                                                   \
        hipLaunchKernelGGL((k_extract_wave<CW, false, CODE, true>), dim3(grid_for(s->n, per_wg)), dim3(256), 0, ctx->stream,      \
                           s->chars, s->offsets, s->n, P, v->alpha_map, s->slots, s->row_cnt, nullptr, per_wg, V, row_nnz)
        if (s->char_width == 1) {
            if (v->code_bits <= 32) PFZ_EXTRACT_ROWS(1, uint32_t);
            else PFZ_EXTRACT_ROWS(1, uint64_t);
        } else {
            if (v->code_bits <= 32) PFZ_EXTRACT_ROWS(4, uint32_t);
            else PFZ_EXTRACT_ROWS(4, uint64_t);
        }
#undef PFZ_EXTRACT_ROWS
        PFZ_HIP(hipGetLastError());
        *rows_done = true;
        return PFZ_OK;
    }
#define PFZ_EXTRACT(CW, LB)                                                                                          \
    do {                                                                                                             \
        if (wave && v->code_bits <= 32)                                                                              \
            hipLaunchKernelGGL((k_extract_wave<CW, LB, uint32_t>), dim3(grid_for(s->n, per_wg) + grid_b), dim3(256), 0,       \
                               ctx->stream, s->chars, s->offsets, s->n, P, v->alpha_map, s->slots, s->row_cnt,       \
                               mark ? v->bitmap : nullptr, per_wg, VocabView{}, nullptr, B);                         \
        else if (wave)                                                                                               \
            hipLaunchKernelGGL((k_extract_wave<CW, LB, uint64_t>), dim3(grid_for(s->n, per_wg) + grid_b), dim3(256), 0,       \
                               ctx->stream, s->chars, s->offsets, s->n, P, v->alpha_map, s->slots, s->row_cnt,       \
                               mark ? v->bitmap : nullptr, per_wg, VocabView{}, nullptr, B);                         \
        else if (v->code_bits <= 32)                                                                                 \
            hipLaunchKernelGGL((k_extract<CW, LB, uint32_t>), dim3(grid_for(s->n)), dim3(256), 0, ctx->stream,       \
                               s->chars, s->offsets, s->n, P, v->alpha_map, s->slots, s->row_cnt,                    \
                               mark ? v->bitmap : nullptr);                                                          \
        else                                                                                                         \
            hipLaunchKernelGGL((k_extract<CW, LB, uint64_t>), dim3(grid_for(s->n)), dim3(256), 0, ctx->stream,       \
                               s->chars, s->offsets, s->n, P, v->alpha_map, s->slots, s->row_cnt,                    \
                               mark ? v->bitmap : nullptr);                                                          \
    } while (0)
    if (s->char_width == 1) {
        if (lds_bitmap) PFZ_EXTRACT(1, true);
        else PFZ_EXTRACT(1, false);
    } else {
        if (lds_bitmap) PFZ_EXTRACT(4, true);
        else PFZ_EXTRACT(4, false);
    }
#undef PFZ_EXTRACT
    PFZ_HIP(hipGetLastError());
    return PFZ_OK;
}

// sb: a second list whose SHORT rows ride in the same launch (df must not count: the LDS-histogram path); its long rows, if
// any, are the caller's (run_rows(sb, .., short_done = true)).
static int run_rows(pfz_ctx *ctx, const pfz_tfidf *v, pfz_strings *s, DfSink df, bool short_done = false, pfz_strings *sb = nullptr)
{
    if (s->n == 0) return sb ? run_rows(ctx, v, sb, df) : PFZ_OK;
    const int R = v->params.ngram_hi - v->params.ngram_lo + 1;
    VocabView V{v->bitmap, v->prefix, v->vcodes, v->vocab};
    int32_t *row_nnz = s->row_cnt + (s->n + 1);
    if (!short_done) {
        ListB B;
        unsigned grid_b = 0;
        if (sb && sb->n > 0) {
            B.off = sb->offsets;
            B.n = sb->n;
            B.slots = sb->slots;
            B.row_cnt = sb->row_cnt;
            B.grid0 = grid_for(s->n, 4);
            grid_b = grid_for(sb->n, 4);
        }
        ProfScope ps(ctx, "k2_rows_short");
        hipLaunchKernelGGL(k_rows_short, dim3(grid_for(s->n, 4) + grid_b), dim3(256), 0, ctx->stream, s->offsets, s->n, R, V,
                           s->slots, s->row_cnt, row_nnz, df, B);
    }
    if (s->max_len * R > kShortMax) {
        const int64_t max_cnt = s->max_len * R;
        const int64_t n_tiles = (s->n + kLongTile - 1) / kLongTile;
        unsigned grid = (unsigned)std::min<int64_t>(n_tiles, 2048);
        uint32_t *giant = nullptr;
        int64_t stride = 0;
        if (max_cnt > kLongMax) {
            grid = (unsigned)std::min<int64_t>(n_tiles, 64);
            stride = 128;
            while (stride < max_cnt) stride <<= 1;
            PFZ_TRY(ensure_scratch(ctx, (size_t)grid * (size_t)stride * sizeof(uint32_t)));
            giant = (uint32_t *)ctx->scratch;
        }
        ProfScope ps(ctx, "k2_rows_long");
        hipLaunchKernelGGL(k_rows_long, dim3(grid), dim3(256), 0, ctx->stream, s->offsets, s->n, R, V, s->slots,
                           s->row_cnt, row_nnz, df, giant, stride);
    }
    PFZ_HIP(hipGetLastError());
    if (sb && !short_done) return run_rows(ctx, v, sb, df, true);      // (the second list's long rows)
    return PFZ_OK;
}

// rank prefix of the vocabulary bitmap; the vocabulary's size starts its way to the host (*size: lazy_get it when needed)
static int build_prefix(pfz_ctx *ctx, pfz_tfidf *v, LazyI32 *size)
{
    if (v->n_groups >= 1 && v->n_groups <= 1024) {      // (beyond: the two-launch path -- wider codes reach it, tests/test_vectorize_gpu.py)
        if (size) PFZ_TRY(lazy_acquire(ctx, size));
        hipLaunchKernelGGL(k_group_popc_scan, dim3(1), dim3(1024), 0, ctx->stream, v->bitmap, (int32_t)v->n_groups, v->prefix,
                           size ? size->slot : nullptr);
        PFZ_HIP(hipGetLastError());
        return size ? lazy_mark(ctx, size) : PFZ_OK;
    }
    hipLaunchKernelGGL(k_group_popc, dim3(grid_for(v->n_groups)), dim3(256), 0, ctx->stream, v->bitmap, v->n_groups,
                       v->prefix);
    return exclusive_scan_i32(ctx, v->prefix, v->n_groups, size);
}

// an upper bound of the vocabulary known before anything ran: the n-grams an alphabet of `a` symbols can form
static int64_t vocab_bound(int64_t a, int lo, int hi)
{
    int64_t total = 0, p = 1;
    for (int n = 1; n <= hi; ++n) {
        if (p > ((int64_t)1 << 40) / (a > 0 ? a : 1)) return (int64_t)1 << 40;
        p *= a;
        if (n >= lo) total += p;
    }
    return total;
}

// Sorted-vocabulary mode (codes wider than kBitmapMaxBits): gather the codes k_extract left in the slot
// ranges of the fitted lists, sort them, keep the distinct ones.  v->vcodes / v->vocab on return.
static int build_sorted_vocab(pfz_ctx *ctx, pfz_tfidf *v, pfz_strings *const lists[2])
{
    const int R = v->params.ngram_hi - v->params.ngram_lo + 1;
    struct Tmp {
        void *p = nullptr;
        ~Tmp() { if (p) pool_free(p); }
    } code_off[2], gathered, sorted, flags;
    int64_t base[2] = {0, 0}, total = 0;
    for (int li = 0; li < 2; ++li) {
        pfz_strings *s = lists[li];
        if (!s || s->n == 0) continue;
        PFZ_TRY(pool_alloc(ctx, &code_off[li].p, (size_t)(s->n + 1) * sizeof(int32_t)));
        hipLaunchKernelGGL(k_copy_i32, dim3(grid_for(s->n)), dim3(256), 0, ctx->stream, s->row_cnt, s->n, (int32_t *)code_off[li].p);
        PFZ_TRY(exclusive_scan_i32(ctx, (int32_t *)code_off[li].p, s->n));
        int32_t t = 0;
        PFZ_HIP(hipMemcpyAsync(&t, (int32_t *)code_off[li].p + s->n, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        PFZ_HIP(hipStreamSynchronize(ctx->stream));
        base[li] = total;
        total += t;
    }
    v->vocab = 0;
    if (total == 0) return PFZ_OK;
    if (total >= ((int64_t)1 << 31) - 2) {
        set_error("pfz_tfidf_fit: %lld n-gram occurrences exceed the int32 layout of the sorted-vocabulary path", (long long)total);
        return PFZ_ERR_UNSUPPORTED;
    }
    PFZ_TRY(pool_alloc(ctx, &gathered.p, (size_t)total * sizeof(uint64_t)));
    PFZ_TRY(pool_alloc(ctx, &sorted.p, (size_t)sort_codes_capacity(total) * sizeof(uint64_t)));
    PFZ_TRY(pool_alloc(ctx, &flags.p, (size_t)(total + 1) * sizeof(int32_t)));
    for (int li = 0; li < 2; ++li) {
        pfz_strings *s = lists[li];
        if (!s || s->n == 0) continue;
        hipLaunchKernelGGL(k_gather_codes, dim3(grid_for(s->n, 16)), dim3(256), 0, ctx->stream, s->offsets, s->n, R, s->slots,
                           (const int32_t *)code_off[li].p, (uint64_t *)gathered.p + base[li]);
    }
    PFZ_TRY(sort_codes_u64(ctx, (const uint64_t *)gathered.p, (uint64_t *)sorted.p, total));
    hipLaunchKernelGGL(k_flag_heads, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const uint64_t *)sorted.p, total,
                       (int32_t *)flags.p);
    PFZ_TRY(exclusive_scan_i32(ctx, (int32_t *)flags.p, total));
    int32_t n_distinct = 0;
    PFZ_HIP(hipMemcpyAsync(&n_distinct, (int32_t *)flags.p + total, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    PFZ_HIP(hipStreamSynchronize(ctx->stream));
    PFZ_TRY(pool_alloc(ctx, &v->vcodes, (size_t)n_distinct * sizeof(uint64_t)));
    hipLaunchKernelGGL(k_scatter_heads, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const uint64_t *)sorted.p, total,
                       (const int32_t *)flags.p, v->vcodes);
    PFZ_HIP(hipGetLastError());
    v->vocab = n_distinct;
    return PFZ_OK;
}

// Sorted-vocabulary mode across ranks: every rank holds the distinct codes of ITS lists; the vocabulary is their union.
// The ranks' arrays differ in length, so the lengths are all-gathered first, the arrays padded with all-ones keys to the
// longest, all-gathered, sorted as one array (the padding sorts to the end and is cut off) and made distinct again --
// every rank ends with the same vcodes a single-GPU fit on the concatenated lists would build.
static int merge_sorted_vocab(pfz_ctx *ctx, pfz_comm *comm, pfz_tfidf *v)
{
    const int world = comm_world(comm);
    struct Tmp {
        void *p = nullptr;
        ~Tmp() { if (p) pool_free(p); }
    } sizes, send, gathered, sorted, flags;
    PFZ_TRY(pool_alloc(ctx, &sizes.p, (size_t)(world + 1) * sizeof(int64_t)));
    const int64_t mine = v->vocab;
    PFZ_TRY(copy_h2d(ctx, (int64_t *)sizes.p + world, &mine, sizeof(int64_t)));
    PFZ_TRY(comm_allgather_bytes(comm, (int64_t *)sizes.p + world, sizes.p, sizeof(int64_t)));
    std::vector<int64_t> h((size_t)world);
    PFZ_TRY(copy_d2h(ctx, h.data(), sizes.p, (size_t)world * sizeof(int64_t)));
    int64_t longest = 0, total = 0;
    for (int64_t x : h) {
        longest = std::max(longest, x);
        total += x;
    }
    Tmp old_codes;                        // this rank's own codes: released when the merge is enqueued
    old_codes.p = v->vcodes;
    v->vcodes = nullptr;
    v->vocab = 0;
    if (total == 0) return PFZ_OK;
    if ((int64_t)world * longest >= ((int64_t)1 << 31) - 2) {
        set_error("pfz_tfidf_fit_sharded: %lld distinct n-grams per rank exceed the int32 layout of the sorted-vocabulary path",
                  (long long)longest);
        return PFZ_ERR_UNSUPPORTED;
    }
    const int64_t padded = (int64_t)world * longest;
    PFZ_TRY(pool_alloc(ctx, &send.p, (size_t)longest * sizeof(uint64_t)));
    PFZ_HIP(hipMemsetAsync(send.p, 0xff, (size_t)longest * sizeof(uint64_t), ctx->stream));
    if (mine > 0) PFZ_HIP(hipMemcpyAsync(send.p, old_codes.p, (size_t)mine * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
    PFZ_TRY(pool_alloc(ctx, &gathered.p, (size_t)padded * sizeof(uint64_t)));
    PFZ_TRY(comm_allgather_bytes(comm, send.p, gathered.p, (size_t)longest * sizeof(uint64_t)));
    PFZ_TRY(pool_alloc(ctx, &sorted.p, (size_t)sort_codes_capacity(padded) * sizeof(uint64_t)));
    PFZ_TRY(pool_alloc(ctx, &flags.p, (size_t)(total + 1) * sizeof(int32_t)));
    PFZ_TRY(sort_codes_u64(ctx, (const uint64_t *)gathered.p, (uint64_t *)sorted.p, padded));
    // (the first `total` keys are the real ones: a real all-ones code would be among them, the padding behind)
    hipLaunchKernelGGL(k_flag_heads, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const uint64_t *)sorted.p, total,
                       (int32_t *)flags.p);
    PFZ_TRY(exclusive_scan_i32(ctx, (int32_t *)flags.p, total));
    int32_t n_distinct = 0;
    PFZ_TRY(copy_d2h(ctx, &n_distinct, (int32_t *)flags.p + total, sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &v->vcodes, (size_t)n_distinct * sizeof(uint64_t)));
    hipLaunchKernelGGL(k_scatter_heads, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const uint64_t *)sorted.p, total,
                       (const int32_t *)flags.p, v->vcodes);
    PFZ_HIP(hipGetLastError());
    v->vocab = n_distinct;
    return PFZ_OK;
}

static int alloc_vocab_space(pfz_ctx *ctx, pfz_tfidf *v)
{
    const int64_t n_bits = std::max<int64_t>((int64_t)1 << v->code_bits, 256);
    v->n_groups = n_bits / 256;
    PFZ_TRY(pool_alloc(ctx, &v->bitmap, (size_t)(n_bits / 8)));
    PFZ_TRY(pool_alloc(ctx, &v->prefix, (size_t)(v->n_groups + 1) * sizeof(int32_t)));
    PFZ_HIP(hipMemsetAsync(v->bitmap, 0, (size_t)(n_bits / 8), ctx->stream));
    return PFZ_OK;
}

static int check_params(const pfz_tfidf_params *p)
{
    PFZ_REQUIRE(p, "pfz_tfidf: params is NULL");
    PFZ_REQUIRE(p->ngram_lo >= 1 && p->ngram_hi >= p->ngram_lo, "pfz_tfidf: bad n_gram_range (%d, %d)", p->ngram_lo,
                p->ngram_hi);
    return PFZ_OK;
}

static int set_alphabet(pfz_ctx *ctx, pfz_tfidf *v, const std::vector<uint32_t> &sorted_cps)
{
    v->alphabet = sorted_cps;
    const uint64_t S = sorted_cps.size();
    v->bits_per_char = bits_for(S > 0 ? S : 1);
    v->code_bits = v->bits_per_char * v->params.ngram_hi;
    if (v->code_bits > kMaxCodeBits) {
        set_error("pfz_tfidf: n_gram_range upper bound %d with an alphabet of %llu symbols needs %d-bit n-gram codes; "
                  "this build supports up to %d bits (e.g. cleaned strings up to 10-grams, 255 symbols up to 8-grams)",
                  v->params.ngram_hi, (unsigned long long)S, v->code_bits, kMaxCodeBits);
        return PFZ_ERR_UNSUPPORTED;
    }
    const uint32_t max_cp = S ? sorted_cps.back() : 0;
    std::vector<uint32_t> map((size_t)max_cp + 1, 0u);
    for (size_t r = 0; r < sorted_cps.size(); ++r) map[sorted_cps[r]] = (uint32_t)r + 1u;
    v->alpha_map_len = (int64_t)map.size();
    PFZ_TRY(pool_alloc(ctx, &v->alpha_map, map.size() * sizeof(uint32_t)));
    PFZ_HIP(hipMemcpyAsync(v->alpha_map, map.data(), map.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    PFZ_HIP(hipStreamSynchronize(ctx->stream));   // `map` is a temporary
    return PFZ_OK;
}

// cleaned alphabet: ' ' 0-9 a-z  -> ranks 1..37
static inline uint32_t clean_rank_to_cp(uint32_t m) { return m == 1 ? ' ' : (m <= 11 ? '0' + (m - 2) : 'a' + (m - 12)); }
static inline uint32_t clean_cp_to_rank(uint32_t c)
{
    if (c == ' ') return 1;
    if (c >= '0' && c <= '9') return c - '0' + 2;
    if (c >= 'a' && c <= 'z') return c - 'a' + 12;
    return 0;
}

}  // namespace pfz

using namespace pfz;

extern "C" {

int pfz_strings_upload(pfz_ctx *ctx, const void *chars, const int64_t *offsets, int64_t n, int32_t char_width,
                       pfz_strings **out)
{
    PFZ_REQUIRE(ctx && out && offsets && n >= 0, "pfz_strings_upload: bad arguments");
    PFZ_REQUIRE(char_width == 1 || char_width == 4, "pfz_strings_upload: char_width must be 1 or 4 (got %d)", char_width);
    PFZ_REQUIRE(offsets[0] == 0, "pfz_strings_upload: offsets[0] must be 0");
    const int64_t n_units = offsets[n];
    PFZ_REQUIRE(n_units >= 0, "pfz_strings_upload: negative total length");
    PFZ_REQUIRE(n_units == 0 || chars, "pfz_strings_upload: chars is NULL");
    if (n >= ((int64_t)1 << 31) - 2 || n_units >= ((int64_t)1 << 40)) {
        set_error("pfz_strings_upload: list too large (%lld strings, %lld code units)", (long long)n, (long long)n_units);
        return PFZ_ERR_UNSUPPORTED;
    }
    PFZ_HIP(hipSetDevice(ctx->device));
    Owner<pfz_strings, pfz_strings_free> s(new pfz_strings());
    s->ctx = ctx;
    s->n = n;
    s->n_units = n_units;
    s->char_width = char_width;
    const size_t char_bytes = (size_t)n_units * (size_t)char_width, off_bytes = (size_t)(n + 1) * sizeof(int64_t);
    if (char_bytes + off_bytes <= (48u << 10)) {
        // a query batch: offsets and code units in ONE block and one host-to-device copy (every copy is ~10 us of a 150-us
        // single-query match); `offsets` is the block, `chars` points into it
        const size_t off_pad = (off_bytes + 255) & ~(size_t)255;
        PFZ_TRY(pool_alloc(ctx, &s->offsets, off_pad + char_bytes + 16));
        s->chars = (char *)s->offsets + off_pad;
        s->chars_in_offsets = true;
        std::vector<char> both(off_pad + char_bytes);
        memcpy(both.data(), offsets, off_bytes);
        if (char_bytes) memcpy(both.data() + off_pad, chars, char_bytes);
        PFZ_TRY(copy_h2d(ctx, s->offsets, both.data(), both.size()));
    } else {
        PFZ_TRY(pool_alloc(ctx, &s->chars, (size_t)(n_units > 0 ? n_units : 1) * (size_t)char_width + 16));
        PFZ_TRY(pool_alloc(ctx, &s->offsets, off_bytes));
        if (n_units > 0) PFZ_TRY(copy_h2d(ctx, s->chars, chars, char_bytes));
        PFZ_TRY(copy_h2d(ctx, s->offsets, offsets, off_bytes));
    }
    // (the host's own look at the offsets BEHIND the copies: the DMA is on its way while 100 000 lengths are checked and kept)
    int64_t max_len = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t len = offsets[i + 1] - offsets[i];
        PFZ_REQUIRE(len >= 0, "pfz_strings_upload: offsets not monotone at %lld", (long long)i);
        if (len > max_len) max_len = len;
    }
    s->max_len = max_len;
    s->h_off.assign(offsets, offsets + n + 1);
    *out = s.release();
    return PFZ_OK;
}

void pfz_strings_free(pfz_strings *s)
{
    if (!s) return;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    if (s->chars && !s->chars_in_offsets) pool_free(s->chars);
    if (s->offsets) pool_free(s->offsets);
    if (s->slots) pool_free(s->slots);
    if (s->indel_plan) pfz_indel_plan_free(s->indel_plan);
    if (s->fuzz_plan) pfz_fuzz_plan_free(s->fuzz_plan);
    if (s->fuzz_forms) pfz_fuzz_forms_free(s->fuzz_forms);
    if (s->row_cnt) pool_free(s->row_cnt);
    delete s;
}

void pfz_tfidf_free(pfz_tfidf *v)
{
    if (!v) return;
    if (v->ctx) (void)hipSetDevice(v->ctx->device);
    if (v->alpha_map) pool_free(v->alpha_map);
    if (v->bitmap) pool_free(v->bitmap);
    if (v->prefix) pool_free(v->prefix);
    if (v->vcodes) pool_free(v->vcodes);
    if (v->df) pool_free(v->df);
    if (v->idf) pool_free(v->idf);
    delete v;
}

}  // extern "C"

// OR `world` bitmaps of n_words words each (gathered back to back) into dst
__global__ __launch_bounds__(256) void k_or_reduce(const uint32_t *__restrict__ gathered, int64_t n_words, int world,
                                                    uint32_t *__restrict__ dst)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * 256) {
        uint32_t v = 0;
        for (int r = 0; r < world; ++r) v |= gathered[(int64_t)r * n_words + i];
        dst[i] = v;
    }
}

// Fit on docs_a (identical on every rank of `comm`, counted once) + docs_b (this
// rank's shard).  comm == NULL: single-GPU fit on docs_a + docs_b.
static int fit_impl(pfz_ctx *ctx, pfz_comm *comm, const pfz_tfidf_params *params, const pfz_strings *docs_a_c,
                    const pfz_strings *docs_b_c, pfz_tfidf **out)
{
    const int world = comm_world(comm), rank = comm_rank(comm);
    PFZ_REQUIRE(ctx && out, "pfz_tfidf_fit: NULL argument");
    PFZ_TRY(check_params(params));
    PFZ_REQUIRE(docs_a_c || docs_b_c, "pfz_tfidf_fit: no documents");
    PFZ_HIP(hipSetDevice(ctx->device));
    // the n-gram cache lives inside the (otherwise read-only) string lists
    pfz_strings *lists[2] = {const_cast<pfz_strings *>(docs_a_c), const_cast<pfz_strings *>(docs_b_c)};
    if (lists[0] == lists[1]) lists[1] = nullptr;
    pfz_tfidf *v = new pfz_tfidf();
    struct Guard {
        pfz_tfidf *p;
        ~Guard() { if (p) pfz_tfidf_free(p); }
    } guard{v};
    v->ctx = ctx;
    v->params = *params;
    v->gen = g_gen.fetch_add(1);

    if (params->clean) {
        for (pfz_strings *s : lists)
            if (s && s->char_width != 1) {
                set_error("pfz_tfidf_fit: clean=1 runs on 1-byte code units only; pre-clean wide strings on the host");
                return PFZ_ERR_INVALID;
            }
        v->bits_per_char = 6;
        v->code_bits = 6 * params->ngram_hi;
        if (v->code_bits > kMaxCodeBits) {
            set_error("pfz_tfidf_fit: n_gram_range upper bound %d needs %d-bit codes; this build supports up to %d-grams "
                      "of cleaned strings", params->ngram_hi, v->code_bits, kMaxCodeBits / 6);
            return PFZ_ERR_UNSUPPORTED;
        }
    } else {
        // alphabet = distinct code units of the fitted documents, in code-point order
        const size_t words = 0x110000 / 32;
        PFZ_TRY(ensure_scratch(ctx, (size_t)(world + 1) * words * sizeof(uint32_t)));
        uint32_t *present = (uint32_t *)ctx->scratch;
        PFZ_HIP(hipMemsetAsync(present, 0, words * sizeof(uint32_t), ctx->stream));
        for (pfz_strings *s : lists) {
            if (!s || s->n_units == 0) continue;
            const unsigned grid = (unsigned)std::min<int64_t>((s->n_units + 255) / 256, 4096);
            if (s->char_width == 1)
                hipLaunchKernelGGL(k_alpha_mark<1>, dim3(grid), dim3(256), 0, ctx->stream, s->chars, s->n_units, present);
            else
                hipLaunchKernelGGL(k_alpha_mark<4>, dim3(grid), dim3(256), 0, ctx->stream, s->chars, s->n_units, present);
        }
        if (world > 1) {   // union of the ranks' alphabets
            uint32_t *gathered = present + words;
            PFZ_TRY(comm_allgather_bytes(comm, present, gathered, words * sizeof(uint32_t)));
            hipLaunchKernelGGL(k_or_reduce, dim3(64), dim3(256), 0, ctx->stream, gathered, (int64_t)words, world, present);
        }
        std::vector<uint32_t> h(words);
        PFZ_HIP(hipMemcpyAsync(h.data(), present, words * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        PFZ_HIP(hipStreamSynchronize(ctx->stream));
        std::vector<uint32_t> cps;
        for (size_t wi = 0; wi < words; ++wi) {
            uint32_t word = h[wi];
            while (word) {
                const int bit = __builtin_ctz(word);
                word &= word - 1;
                cps.push_back((uint32_t)(wi * 32 + bit));
            }
        }
        PFZ_TRY(set_alphabet(ctx, v, cps));
    }
    const bool sorted_vocab = v->code_bits > kBitmapMaxBits;
    if (sorted_vocab) {
        for (pfz_strings *s : lists)
            if (s) PFZ_TRY(run_extract(ctx, v, s, false));
        PFZ_TRY(build_sorted_vocab(ctx, v, lists));
        if (world > 1) PFZ_TRY(merge_sorted_vocab(ctx, comm, v));      // vocabulary = union of the ranks' distinct codes
    } else {
        PFZ_TRY(alloc_vocab_space(ctx, v));
        bool b_done = false;
        if (lists[0]) PFZ_TRY(run_extract(ctx, v, lists[0], true, nullptr, world == 1 ? lists[1] : nullptr, &b_done));
        if (lists[1] && !b_done) PFZ_TRY(run_extract(ctx, v, lists[1], true));
    }
    if (world > 1 && !sorted_vocab) {   // vocabulary = union of the ranks' n-gram sets
        const int64_t n_words = v->n_groups * 8;
        if (v->code_bits > 30) {
            set_error("pfz_tfidf_fit_sharded: %d-bit n-gram codes: the bitmap all-gather is limited to 30 bits", v->code_bits);
            return PFZ_ERR_UNSUPPORTED;
        }
        PFZ_TRY(ensure_scratch(ctx, (size_t)world * (size_t)n_words * sizeof(uint32_t)));
        uint32_t *gathered = (uint32_t *)ctx->scratch;
        PFZ_TRY(comm_allgather_bytes(comm, v->bitmap, gathered, (size_t)n_words * sizeof(uint32_t)));
        const unsigned grid = (unsigned)std::min<int64_t>((n_words + 255) / 256, 4096);
        hipLaunchKernelGGL(k_or_reduce, dim3(grid), dim3(256), 0, ctx->stream, gathered, n_words, world, v->bitmap);
    }
    // The vocabulary's size is needed on the host (allocation sizes below), but not by the row kernels: where the alphabet bounds
    // it inside what one LDS histogram holds -- cleaned 3-grams: 37^3 -- the rows of the fitted lists are sorted and counted
    // (k_rows_*) while the size is on its way, instead of after a synchronising read-back (45 us of idle device per fit).
    LazyI32 vsize;
    bool rows_done = false;
    if (!sorted_vocab) {
        PFZ_TRY(build_prefix(ctx, v, &vsize));
        const int64_t a_syms = params->clean ? 37 : (int64_t)v->alphabet.size() + 1;
        const bool small = vocab_bound(a_syms, params->ngram_lo, params->ngram_hi) <= 2 * (int64_t)kHistWords && !getenv("PFZ_NO_LDS_HIST");
        int rc = PFZ_OK;
        if (small && world == 1) {
            const char *two_knob = getenv("PFZ_K1_TWO_LISTS");
            if (lists[0] && lists[1] && !(two_knob && atoi(two_knob) == 0)) {
                rc = run_rows(ctx, v, lists[0], DfSink{nullptr, 0}, false, lists[1]);
            } else {
                for (pfz_strings *s : lists)
                    if (s && rc == PFZ_OK) rc = run_rows(ctx, v, s, DfSink{nullptr, 0});
            }
            rows_done = true;
        }
        int32_t total = 0;
        const int rc2 = lazy_get(ctx, &vsize, &total);      // (also on the error path: the slot goes back to the pool)
        PFZ_TRY(rc);
        PFZ_TRY(rc2);
        v->vocab = total;
    }
    if (v->vocab == 0) {
        // sklearn text.py:1282-1285
        set_error("empty vocabulary; perhaps the documents only contain stop words");
        return PFZ_ERR_INVALID;
    }
    PFZ_TRY(pool_alloc(ctx, &v->df, (size_t)v->vocab * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &v->idf, (size_t)v->vocab * sizeof(double)));
    // document frequencies: LDS histograms per kHistRows strings when the vocabulary fits one
    // (every realistic case), else sharded global counters -- up to 32 per n-gram, at most 64 Mi in total
    const bool lds_hist = v->vocab <= 2 * (int64_t)kHistWords && !getenv("PFZ_NO_LDS_HIST");   // env: tests
    int df_shift = 5;
    while (df_shift > 0 && (v->vocab << df_shift) > ((int64_t)64 << 20)) --df_shift;
    const int32_t words = (int32_t)((v->vocab + 1) / 2);
    int64_t chunks = 0;
    for (int li = 0; li < 2; ++li)
        if (lists[li] && ((li == 1) || rank == 0 || world == 1)) chunks += (lists[li]->n + kHistRows - 1) / kHistRows;
    int32_t *df_sh = nullptr;   // sharded counters, or the partial histograms
    const size_t df_sh_bytes = lds_hist ? (size_t)(chunks > 0 ? chunks : 1) * (size_t)words * sizeof(uint32_t)
                                        : (size_t)(v->vocab << df_shift) * sizeof(int32_t);
    PFZ_TRY(pool_alloc(ctx, &df_sh, df_sh_bytes));
    struct ShGuard {
        int32_t *p;
        ~ShGuard() { pool_free(p); }
    } sh_guard{df_sh};
    if (!lds_hist) PFZ_HIP(hipMemsetAsync(df_sh, 0, df_sh_bytes, ctx->stream));
    v->n_docs = 0;
    int64_t local_docs = 0, chunk0 = 0;
    const int R = v->params.ngram_hi - v->params.ngram_lo + 1;
    // (both lists' histograms in one launch where both count and the rows are done: the second list's chunks follow the first's)
    const char *two_knob = getenv("PFZ_K1_TWO_LISTS");
    const bool hist_both = world == 1 && rows_done && lds_hist && lists[0] && lists[1] && lists[0]->n > 0 && lists[1]->n > 0 &&
                           !(two_knob && atoi(two_knob) == 0);
    if (hist_both) {
        pfz_strings *a = lists[0], *b = lists[1];
        const int64_t nch_a = (a->n + kHistRows - 1) / kHistRows, nch_b = (b->n + kHistRows - 1) / kHistRows;
        ListB B;
        B.off = b->offsets;
        B.n = b->n;
        B.slots = b->slots;
        B.row_cnt = b->row_cnt;
        B.grid0 = (unsigned)nch_a;
        ProfScope ps(ctx, "k2_df_hist");
        hipLaunchKernelGGL(k_df_hist, dim3((unsigned)(nch_a + nch_b)), dim3(1024), 0, ctx->stream, a->offsets, a->n, R, a->slots,
                           a->row_cnt + (a->n + 1), words, (uint32_t *)df_sh, B);
    }
    for (int li = 0; li < 2; ++li) {
        pfz_strings *s = lists[li];
        if (!s) continue;
        // the replicated list (docs_a) is counted by rank 0 only
        const bool counts = (li == 1) || rank == 0 || world == 1;
        if (!rows_done) PFZ_TRY(run_rows(ctx, v, s, DfSink{counts && !lds_hist ? df_sh : nullptr, df_shift}));
        s->cache_gen = v->gen;
        if (counts) local_docs += s->n;
        if (counts && lds_hist && s->n > 0 && !hist_both) {
            const int64_t nch = (s->n + kHistRows - 1) / kHistRows;
            ProfScope ps(ctx, "k2_df_hist");
            hipLaunchKernelGGL(k_df_hist, dim3((unsigned)nch), dim3(1024), 0, ctx->stream, s->offsets, s->n, R, s->slots,
                               s->row_cnt + (s->n + 1), words, (uint32_t *)df_sh + chunk0 * words);
            chunk0 += nch;
        }
    }
    const bool idf_fused = lds_hist && world == 1;
    if (lds_hist) {
        ProfScope ps(ctx, "k2_df_hist");
        hipLaunchKernelGGL(k_df_hist_reduce, dim3(grid_for(words)), dim3(256), 0, ctx->stream, (const uint32_t *)df_sh, words,
                           (int32_t)chunks, v->vocab, v->df, (double)local_docs, idf_fused ? v->idf : nullptr);
    } else {
        hipLaunchKernelGGL(k_df_reduce, dim3(grid_for(v->vocab)), dim3(256), 0, ctx->stream, df_sh, v->vocab, df_shift, v->df);
    }
    v->n_docs = local_docs;
    if (world > 1) {
        PFZ_TRY(comm_allreduce_sum_i32(comm, v->df, (size_t)v->vocab));
        int64_t *d_n = nullptr;
        PFZ_TRY(pool_alloc(ctx, &d_n, sizeof(int64_t)));
        hipError_t e = hipMemcpyAsync(d_n, &local_docs, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
        int rc = PFZ_OK;
        if (e == hipSuccess) rc = comm_allreduce_sum_i64(comm, d_n, 1);
        int64_t total = 0;
        if (e == hipSuccess && rc == PFZ_OK) e = hipMemcpyAsync(&total, d_n, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rc == PFZ_OK) e = hipStreamSynchronize(ctx->stream);
        pool_free(d_n);
        PFZ_TRY(rc);
        PFZ_HIP(e);
        v->n_docs = total;
    }
    if (!idf_fused)
        hipLaunchKernelGGL(k_idf, dim3(grid_for(v->vocab)), dim3(256), 0, ctx->stream, v->df, v->vocab, (double)v->n_docs,
                           v->idf);
    PFZ_HIP(hipGetLastError());
    guard.p = nullptr;
    *out = v;
    return PFZ_OK;
}

extern "C" {

int pfz_tfidf_fit(pfz_ctx *ctx, const pfz_tfidf_params *params, const pfz_strings *docs_a, const pfz_strings *docs_b,
                  pfz_tfidf **out)
{
    return fit_impl(ctx, nullptr, params, docs_a, docs_b, out);
}

int pfz_tfidf_fit_sharded(pfz_ctx *ctx, pfz_comm *comm, const pfz_tfidf_params *params, const pfz_strings *replicated,
                          const pfz_strings *local_shard, pfz_tfidf **out)
{
    PFZ_REQUIRE(comm, "pfz_tfidf_fit_sharded: NULL communicator");
    PFZ_REQUIRE(replicated != local_shard || !replicated, "pfz_tfidf_fit_sharded: the replicated and the sharded list must differ");
    return fit_impl(ctx, comm, params, replicated, local_shard, out);
}

int pfz_tfidf_info(const pfz_tfidf *v, int64_t *vocab_size, int64_t *n_docs, int32_t *code_bits)
{
    PFZ_REQUIRE(v, "pfz_tfidf_info: NULL vectoriser");
    if (vocab_size) *vocab_size = v->vocab;
    if (n_docs) *n_docs = v->n_docs;
    if (code_bits) *code_bits = v->code_bits;
    return PFZ_OK;
}

int pfz_tfidf_transform(pfz_ctx *ctx, const pfz_tfidf *v, const pfz_strings *docs_c, pfz_csr **out)
{
    PFZ_REQUIRE(ctx && v && docs_c && out, "pfz_tfidf_transform: NULL argument");
    PFZ_HIP(hipSetDevice(ctx->device));
    pfz_strings *s = const_cast<pfz_strings *>(docs_c);
    if (v->params.clean && s->char_width != 1) {
        set_error("pfz_tfidf_transform: clean=1 runs on 1-byte code units only; pre-clean wide strings on the host");
        return PFZ_ERR_INVALID;
    }
    if (s->cache_gen != v->gen) {
        bool rows_done = false;
        PFZ_TRY(run_extract(ctx, v, s, false, &rows_done));
        PFZ_TRY(run_rows(ctx, v, s, DfSink{nullptr, 0}, rows_done));
        s->cache_gen = v->gen;
    }
    const int R = v->params.ngram_hi - v->params.ngram_lo + 1;
    Owner<pfz_csr, pfz_csr_free> m(new pfz_csr());
    m->ctx = ctx;
    m->n_rows = s->n;
    m->n_cols = v->vocab;
    PFZ_TRY(pool_alloc(ctx, &m->indptr, (size_t)(s->n + 1) * sizeof(int32_t)));
    // No read-back of the number of non-zeros: every entry is a distinct n-gram of its string, so the n-gram slots the list
    // owns (R per code unit) bound it -- indices / data are sized by the bound, the count itself travels to the host behind
    // the scan and is waited for only by whoever asks (csr_nnz: pfz_csr_shape, a download).  A synchronising read-back here
    // idles the device until the host has enqueued the next launch: 35 - 45 us per transform in a 0.37-ms step of 10k x 10k.
    const int64_t cap = std::max<int64_t>(1, std::min<int64_t>(s->n_units * (int64_t)R, ((int64_t)1 << 31) - 1));
    m->nnz_cap = cap;
    PFZ_TRY(pool_alloc(ctx, &m->indices, (size_t)cap * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &m->data, (size_t)cap * sizeof(float)));
    if (s->n > 0) {
        const int32_t *row_nnz = s->row_cnt + (s->n + 1);
        const char *knob = getenv("PFZ_K2_SELF_SCAN");      // (tests: 0 = the long-list path on short lists too)
        const bool no_self_scan = knob && atoi(knob) == 0;
        if (s->n <= kSelfScanRows && !no_self_scan) {
            PFZ_TRY(lazy_acquire(ctx, &m->nnz_lazy));
            {
                ProfScope ps(ctx, "k2_finalize");
                hipLaunchKernelGGL(k_finalize<true>, dim3(grid_for(s->n, 16)), dim3(256), 0, ctx->stream, s->offsets, s->n, R, s->slots,
                                   m->indptr, v->idf, m->indices, m->data, row_nnz, m->nnz_lazy.slot);
            }
            PFZ_TRY(lazy_mark(ctx, &m->nnz_lazy));
        } else {
            hipLaunchKernelGGL(k_copy_i32, dim3(grid_for(s->n)), dim3(256), 0, ctx->stream, row_nnz, s->n, m->indptr);
            PFZ_TRY(exclusive_scan_i32(ctx, m->indptr, s->n, &m->nnz_lazy));
            ProfScope ps(ctx, "k2_finalize");
            hipLaunchKernelGGL(k_finalize<false>, dim3(grid_for(s->n, 16)), dim3(256), 0, ctx->stream, s->offsets, s->n, R, s->slots,
                               m->indptr, v->idf, m->indices, m->data, nullptr, nullptr);
        }
    } else {
        PFZ_HIP(hipMemsetAsync(m->indptr, 0, sizeof(int32_t), ctx->stream));
        m->nnz = 0;
    }
    PFZ_HIP(hipGetLastError());
    *out = m.release();
    return PFZ_OK;
}

int pfz_tfidf_export(pfz_ctx *ctx, const pfz_tfidf *v, uint32_t *ngrams, double *idf, int64_t *df)
{
    PFZ_REQUIRE(ctx && v, "pfz_tfidf_export: NULL argument");
    PFZ_HIP(hipSetDevice(ctx->device));
    PFZ_HIP(hipStreamSynchronize(ctx->stream));
    if (idf && v->vocab > 0) PFZ_HIP(hipMemcpy(idf, v->idf, (size_t)v->vocab * sizeof(double), hipMemcpyDeviceToHost));
    if (df && v->vocab > 0) {
        std::vector<int32_t> h((size_t)v->vocab);
        PFZ_HIP(hipMemcpy(h.data(), v->df, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); ++i) df[i] = h[i];
    }
    if (ngrams && v->vocab > 0) {
        std::vector<uint64_t> codes((size_t)v->vocab);
        if (v->vcodes) {
            PFZ_HIP(hipMemcpy(codes.data(), v->vcodes, codes.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        } else {
            uint64_t *d_codes = nullptr;
            PFZ_TRY(pool_alloc(ctx, &d_codes, (size_t)v->vocab * sizeof(uint64_t)));
            hipLaunchKernelGGL(k_export_codes, dim3(grid_for(v->n_groups)), dim3(256), 0, ctx->stream, v->bitmap, v->prefix,
                               v->n_groups, d_codes);
            hipError_t e = hipMemcpyAsync(codes.data(), d_codes, codes.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            pool_free(d_codes);
            PFZ_HIP(e);
        }
        const int hi = v->params.ngram_hi, w = v->bits_per_char;
        const uint64_t cmask = (1ull << w) - 1ull;
        for (size_t i = 0; i < codes.size(); ++i)
            for (int p = 0; p < hi; ++p) {
                const uint32_t m = (uint32_t)((codes[i] >> ((hi - 1 - p) * w)) & cmask);
                uint32_t cp = 0;
                if (m != 0) cp = v->params.clean ? clean_rank_to_cp(m) : v->alphabet[m - 1];
                ngrams[i * hi + p] = cp;
            }
    }
    return PFZ_OK;
}

int pfz_tfidf_import(pfz_ctx *ctx, const pfz_tfidf_params *params, int64_t vocab, int64_t n_docs, const uint32_t *ngrams,
                     const double *idf, pfz_tfidf **out)
{
    PFZ_REQUIRE(ctx && out && ngrams && idf && vocab > 0, "pfz_tfidf_import: bad arguments");
    PFZ_TRY(check_params(params));
    PFZ_HIP(hipSetDevice(ctx->device));
    pfz_tfidf *v = new pfz_tfidf();
    struct Guard {
        pfz_tfidf *p;
        ~Guard() { if (p) pfz_tfidf_free(p); }
    } guard{v};
    v->ctx = ctx;
    v->params = *params;
    v->gen = g_gen.fetch_add(1);
    v->n_docs = n_docs;
    const int hi = params->ngram_hi;
    if (params->clean) {
        v->bits_per_char = 6;
        v->code_bits = 6 * hi;
        if (v->code_bits > kMaxCodeBits) {
            set_error("pfz_tfidf_import: %d-grams of cleaned strings exceed the %d-bit code space", hi, kMaxCodeBits);
            return PFZ_ERR_UNSUPPORTED;
        }
    } else {
        std::vector<uint32_t> cps;
        for (int64_t i = 0; i < vocab * hi; ++i)
            if (ngrams[i]) cps.push_back(ngrams[i]);
        std::sort(cps.begin(), cps.end());
        cps.erase(std::unique(cps.begin(), cps.end()), cps.end());
        PFZ_TRY(set_alphabet(ctx, v, cps));
    }
    const int w = v->bits_per_char;
    std::vector<uint64_t> codes((size_t)vocab);
    for (int64_t i = 0; i < vocab; ++i) {
        uint64_t code = 0;
        for (int p = 0; p < hi; ++p) {
            const uint32_t cp = ngrams[i * hi + p];
            uint32_t m = 0;
            if (cp != 0) {
                if (params->clean) {
                    m = clean_cp_to_rank(cp);
                } else {
                    auto it = std::lower_bound(v->alphabet.begin(), v->alphabet.end(), cp);
                    m = (uint32_t)(it - v->alphabet.begin()) + 1u;
                }
                PFZ_REQUIRE(m != 0, "pfz_tfidf_import: n-gram %lld has a character outside the cleaned alphabet", (long long)i);
            }
            code = (code << w) | m;
        }
        PFZ_REQUIRE(i == 0 || code > codes[(size_t)i - 1], "pfz_tfidf_import: vocabulary not in sorted order at %lld", (long long)i);
        codes[(size_t)i] = code;
    }
    if (v->code_bits > kBitmapMaxBits) {     // sorted-vocabulary mode: the (ascending) codes ARE the vocabulary
        PFZ_TRY(pool_alloc(ctx, &v->vcodes, (size_t)vocab * sizeof(uint64_t)));
        PFZ_HIP(hipMemcpyAsync(v->vcodes, codes.data(), (size_t)vocab * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        PFZ_HIP(hipStreamSynchronize(ctx->stream));
        v->vocab = vocab;
    } else {
        PFZ_TRY(alloc_vocab_space(ctx, v));
        uint64_t *d_codes = nullptr;
        PFZ_TRY(pool_alloc(ctx, &d_codes, (size_t)vocab * sizeof(uint64_t)));
        hipError_t e = hipMemcpyAsync(d_codes, codes.data(), (size_t)vocab * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_set_bits, dim3(grid_for(vocab)), dim3(256), 0, ctx->stream, d_codes, vocab, v->bitmap);
            e = hipStreamSynchronize(ctx->stream);
        }
        pool_free(d_codes);
        PFZ_HIP(e);
        LazyI32 vsize;
        PFZ_TRY(build_prefix(ctx, v, &vsize));
        int32_t total = 0;
        PFZ_TRY(lazy_get(ctx, &vsize, &total));
        v->vocab = total;
    }
    PFZ_REQUIRE(v->vocab == vocab, "pfz_tfidf_import: %lld distinct n-grams, expected %lld", (long long)v->vocab, (long long)vocab);
    PFZ_TRY(pool_alloc(ctx, &v->df, (size_t)vocab * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &v->idf, (size_t)vocab * sizeof(double)));
    PFZ_HIP(hipMemsetAsync(v->df, 0, (size_t)vocab * sizeof(int32_t), ctx->stream));
    PFZ_HIP(hipMemcpyAsync(v->idf, idf, (size_t)vocab * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PFZ_HIP(hipStreamSynchronize(ctx->stream));
    guard.p = nullptr;
    *out = v;
    return PFZ_OK;
}

}  // extern "C"
