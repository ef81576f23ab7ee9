// K7: the rapidfuzz.fuzz scorers that build a different string pair for every (from, to) -- partial_ratio,
// token_set_ratio, token_ratio, partial_token_*_ratio and WRatio, the default scorer of the reference's RapidFuzz
// matcher (polyfuzz/models/_rapidfuzz.py:45-58, 106-108) -- all pairs + process.extractOne's first best choice.
// (ratio / QRatio / token_sort_ratio are one fixed string per list element: K4, k4_indel.hip.)
//
// Everything is the Indel similarity of SOME pair of strings, and every such pair is a sub-string / sub-token-set of
// three per-string forms the host prepares once per list (polyfuzz_amd/models/_rapidfuzz.py):
//   form 0  the string;   form 1  its whitespace tokens sorted and joined by one space (token_sort);
//   form 2  its DISTINCT tokens sorted and joined (token_set), plus their global ids, lengths.
// As in K4 the from-string is stationary: a workgroup holds the bit-parallel match tables of its from-string's three
// forms in LDS (64-bit words, W words per form) and every lane scores one to-string, lists sorted by length and
// stored [position][lane].  On top of K4's recurrence  V' = (V + (V & PM)) | (V & ~PM):
//   * a SUB-RANGE of the from-form is matched by masking PM (bits below the range never match, the adder's carry
//     chain starts at the range) and the LCS against every PREFIX of it is the number of zero bits of V below the
//     prefix length -- so one pass over the to-string scores all windows of the from-string that start at one
//     position (partial_ratio, from-string the longer one);
//   * windows of the TO-string are separate passes over its characters, the prefixes falling out of the first pass
//     step by step (partial_ratio, from-string the shorter one);
//   * token_set's "tokens of a not in b" / "tokens of b not in a": a mask over the from-form's token positions
//     and a per-character skip on the to-side (a tag per character: token number, separating space or not).
// Scores are float64 with rapidfuzz's two normalisations kept apart: (1 - dist / lensum) * 100 for ratio-like
// values, 100 - 100 dist / lensum inside token_set_ratio.  Windows are compared as exact rationals.
//
// Limits (loud): every form of every from-string <= 256 characters (one, two or four 64-bit words), <= 32 distinct tokens per
// string, alphabet x forms x words within 60 KiB of LDS.   PARITY UNPINNED (rapidfuzz is not installable): the
// oracle is oracle/fuzz_scorers.py, anchored on rapidfuzz's published values.
#include "pfz_internal.h"

#include <algorithm>
#include <climits>
#include <numeric>

namespace pfz {

enum FuzzMode { kWRatio = 0, kPartialRatio = 1, kTokenSetRatio = 2, kTokenRatio = 3, kPartialTokenSortRatio = 4,
                kPartialTokenSetRatio = 5, kPartialTokenRatio = 6 };

struct FuzzArgs {
    // from side (CSR as given by the host)
    const uint16_t *a_sym[3];
    const int64_t *a_off[3];
    const int32_t *a_tok_id, *a_tok_len;
    const int64_t *a_tok_off;
    const int32_t *rows;         // from-rows of this word class
    int32_t n_rows;
    // to side, groups of 64 strings sorted by the length of form 0, [position][lane]
    const uint16_t *b_sym[3];
    const int64_t *b_goff[3];    // [n_groups] element offset of the group in b_sym[v] (and b_tag for v = 2)
    const int32_t *b_gmax[3];    // [n_groups] longest form v in the group
    const int32_t *b_len[3];     // [n_groups * 64]
    const uint8_t *b_tag;        // form 2: token number (5 bits) | 0x80 for the space that follows that token
    const int32_t *b_orig;       // [n_groups * 64] original index, -1 = padding lane
    const int32_t *b_ntok;       // [n_groups * 64]
    const int64_t *b_tgoff;      // [n_groups] element offset of the group's token arrays
    const int32_t *b_tgmax;      // [n_groups] most tokens in the group
    const int32_t *b_tok_id, *b_tok_len;      // [token number][lane]
    int32_t n_groups, n_sym1, mode;
    const int32_t *skip_idx;     // [n_from] or NULL
    int32_t parts;
    double *part_score;          // [n_rows * parts] (parts > 1)
    int32_t *part_idx;
    int32_t *out_idx;            // [n_from]
    double *out_score;
};

__device__ inline double ratio_of(int lcs, int lensum)
{
    const int dist = lensum - 2 * lcs;
    const double norm_dist = lensum != 0 ? (double)dist / (double)lensum : 0.0;
    return (1.0 - norm_dist) * 100.0;
}

__device__ inline double norm_distance(int dist, int lensum)
{
    return lensum != 0 ? 100.0 - (double)(100 * dist) / (double)lensum : 100.0;
}

template <int W>
__device__ inline void bv_step(uint64_t (&V)[W], const uint64_t *pm, const uint64_t (&mask)[W])
{
    uint64_t carry = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint64_t u = V[w] & pm[w] & mask[w];
        const uint64_t sum = V[w] + u + carry;
        carry = (sum < V[w]) | (carry & (sum == V[w]));
        V[w] = sum | (V[w] ^ u);
    }
}

// zero bits of V in positions [0, k)
template <int W>
__device__ inline int zeros_below(const uint64_t (&V)[W], int k)
{
    int n = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int bits = min(max(k - 64 * w, 0), 64);
        const uint64_t m = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
        n += __popcll(~V[w] & m);
    }
    return n;
}

template <int W>
__device__ inline void range_mask(uint64_t (&m)[W], int lo, int hi)      // bits [lo, hi)
{
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int a = min(max(lo - 64 * w, 0), 64), b = min(max(hi - 64 * w, 0), 64);
        const uint64_t below_b = b >= 64 ? ~0ull : ((1ull << b) - 1ull), below_a = a >= 64 ? ~0ull : ((1ull << a) - 1ull);
        m[w] = below_b & ~below_a;
    }
}

template <int W>
__global__ __launch_bounds__(256) void k7_fuzz_kernel(FuzzArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t *pm = (uint64_t *)smem_raw;               // [symbol][form][word]
    __shared__ int s_la[3], s_ta;
    __shared__ int s_tid[32], s_tlen[32];
    __shared__ uint64_t s_tmask[32][W], s_smask[32][W];
    __shared__ double red_s[4];
    __shared__ int red_i[4];
    __shared__ unsigned long long s_best;          // bits of the best score any lane of the workgroup holds so far (>= 0)
    // the to-characters of the form a lane is working on, [position][lane] per wave (groups of up to 64 positions):
    // the window sweeps re-read them |from| times, and a global load per recurrence step is a dependent ~500-cycle
    // round trip (waves sat in s_waitcnt two thirds of their cycles)
    __shared__ uint16_t s_sym[4][64][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mode = A.mode, parts = A.parts;

    for (int p = tid; p < A.n_sym1 * 3 * W; p += 256) pm[p] = 0ull;
    __syncthreads();

    for (int u = blockIdx.x; u < A.n_rows * parts; u += gridDim.x) {
        const int r = u / parts, part = u - r * parts;
        const int row = A.rows[r];
        // ---- the from-string's tables
        for (int v = 0; v < 3; ++v) {
            const int64_t a0 = A.a_off[v][row];
            const int m = (int)(A.a_off[v][row + 1] - a0);
            if (tid == 0) s_la[v] = m;
            for (int p = tid; p < m; p += 256) {
                const int sy = A.a_sym[v][a0 + p];
                if (sy) atomicOr((unsigned long long *)&pm[(sy * 3 + v) * W + (p >> 6)], 1ull << (p & 63));
            }
        }
        if (tid == 0) {
            s_best = 0ull;
            const int64_t t0 = A.a_tok_off[row];
            const int ta = (int)(A.a_tok_off[row + 1] - t0);
            s_ta = ta;
            int start = 0;
            for (int i = 0; i < ta; ++i) {
                const int len = A.a_tok_len[t0 + i];
                s_tid[i] = A.a_tok_id[t0 + i];
                s_tlen[i] = len;
                uint64_t tm[W], sm[W];
                range_mask<W>(tm, start, start + len);
                range_mask<W>(sm, start + len, i + 1 < ta ? start + len + 1 : start + len);
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    s_tmask[i][w] = tm[w];
                    s_smask[i][w] = sm[w];
                }
                start += len + 1;
            }
        }
        __syncthreads();
        const int la0 = s_la[0], la1 = s_la[1], la2 = s_la[2], ta = s_ta;
        const int skip = A.skip_idx ? A.skip_idx[row] : -1;
        double best_score = -1.0;
        int best_idx = INT_MAX;

        // WRatio and the partial_* scorers: the groups whose lengths are within a factor 1.5 of the from-string first
        // -- that is where WRatio's scores above 90 live -- then the others, where whole components can be left out
        // once the workgroup holds a score they cannot reach (see `cur` below).  The token scorers: one pass.
        const bool pruning = mode == kWRatio || mode == kPartialRatio || mode >= kPartialTokenSortRatio;
        const int n_pass = pruning ? 2 : 1;
        for (int pass = 0; pass < n_pass; ++pass)
        for (int g = wave + 4 * part; g < A.n_groups; g += 4 * parts) {
            if (pruning) {
                const int lbmin = __builtin_amdgcn_readfirstlane(A.b_len[0][g * 64]);       // lanes are sorted by length
                const int lbmax = __builtin_amdgcn_readfirstlane(A.b_gmax[0][g]);
                const bool near_group = 3 * lbmax > 2 * la0 && 2 * lbmin < 3 * la0;
                if (near_group != (pass == 0)) continue;
            }
            const double cur = __longlong_as_double((long long)*(volatile unsigned long long *)&s_best);
            const int slot = g * 64 + lane;
            const int orig = A.b_orig[slot];
            const int lb0 = A.b_len[0][slot], lb1 = A.b_len[1][slot], lb2 = A.b_len[2][slot], tb = A.b_ntok[slot];
            uint64_t all[W];
#pragma unroll
            for (int w = 0; w < W; ++w) all[w] = ~0ull;

            int staged = -1;            // (per lane: a lane stages and reads only its own column)
            auto stage = [&](int v, int steps, int64_t off) {
                if (steps > 64 || staged == v) return;
#pragma unroll 4
                for (int pos = 0; pos < steps; ++pos) s_sym[wave][pos][lane] = A.b_sym[v][off + (int64_t)pos * 64];
                staged = v;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            };
            auto sym_at = [&](int v, int steps, int64_t off, int pos) -> int {
                return steps <= 64 ? (int)s_sym[wave][pos][lane] : (int)A.b_sym[v][off + (int64_t)pos * 64];
            };

            // LCS of the from-form v (restricted to `amask`) and the lane's to-form v; with `rb` only the to-tokens
            // whose bit is set are fed (form 2), the space after the last of them left out
            auto lcs_pass = [&](int v, const uint64_t (&amask)[W], bool tagged, uint32_t rb, int last_rb, uint64_t (&V)[W]) {
#pragma unroll
                for (int w = 0; w < W; ++w) V[w] = ~0ull;
                const int steps = __builtin_amdgcn_readfirstlane(A.b_gmax[v][g]);
                const int64_t off = A.b_goff[v][g] + lane;
                for (int pos = 0; pos < steps; ++pos) {        // (one pass: straight from global memory, staging would cost more)
                    int sy = A.b_sym[v][off + (int64_t)pos * 64];
                    if (tagged) {
                        const int tag = A.b_tag[off + (int64_t)pos * 64], j = tag & 31;
                        const bool keep = ((rb >> j) & 1u) && !((tag & 0x80) && j == last_rb);
                        sy = keep ? sy : 0;
                    }
                    bv_step<W>(V, pm + (sy * 3 + v) * W, amask);
                }
            };

            // rapidfuzz.fuzz.partial_ratio of the two v-forms
            auto partial = [&](int v, int la, int lb) -> double {
                if (la == 0 || lb == 0) return la == 0 && lb == 0 ? 100.0 : 0.0;
                int bl = 0, bs = 1;                                     // best window: lcs / (|shorter| + |window|)
                auto cand = [&](int lcs, int sum) {
                    if ((int64_t)lcs * bs > (int64_t)bl * sum) {
                        bl = lcs;
                        bs = sum;
                    }
                };
                const int steps = __builtin_amdgcn_readfirstlane(A.b_gmax[v][g]);
                const int64_t off = A.b_goff[v][g] + lane;
                stage(v, steps, off);
                uint64_t V[W];
                if (__any(lb >= la)) {
                    // the from-form is the shorter (or equal): windows of the to-form starting at s
                    for (int s = 0; s < steps; ++s) {
                        const bool on = lb >= la && s < lb;
                        if (!__any(on)) break;
                        const int wlen = min(la, lb - s);
#pragma unroll
                        for (int w = 0; w < W; ++w) V[w] = ~0ull;
                        for (int k = 0; k < la; ++k) {
                            const int pos = s + k;
                            const int sy = (on && k < wlen && pos < steps) ? sym_at(v, steps, off, pos) : 0;
                            bv_step<W>(V, pm + (sy * 3 + v) * W, all);
                            if (s == 0 && on && k + 1 < la) cand(zeros_below<W>(V, la), la + k + 1);       // prefixes
                        }
                        if (on) cand(zeros_below<W>(V, la), la + wlen);
                    }
                }
                if (__any(lb <= la)) {
                    // the from-form is the longer (or equal): windows of the from-form starting at i, one pass each
                    const bool on = lb <= la;
                    for (int i = 0; i < la; ++i) {
                        uint64_t m[W];
                        range_mask<W>(m, i, la);
#pragma unroll
                        for (int w = 0; w < W; ++w) V[w] = ~0ull;
                        for (int pos = 0; pos < steps; ++pos) {
                            const int sy = on ? sym_at(v, steps, off, pos) : 0;
                            bv_step<W>(V, pm + (sy * 3 + v) * W, m);
                        }
                        if (on) {
                            const int wlen = min(lb, la - i);
                            cand(zeros_below<W>(V, i + wlen) - zeros_below<W>(V, i), lb + wlen);
                            if (i == 0)
                                for (int k = 1; k < lb; ++k) cand(zeros_below<W>(V, k), lb + k);          // prefixes
                        }
                    }
                }
                return ratio_of(bl, bs);
            };

            // common distinct tokens: bit i of ca (from-tokens), bit j of cb (to-tokens)
            uint32_t ca = 0, cb = 0;
            auto intersect = [&]() {
                const int tmax = __builtin_amdgcn_readfirstlane(A.b_tgmax[g]);
                const int64_t toff = A.b_tgoff[g] + lane;
                for (int j = 0; j < tmax; ++j) {
                    const int idb = j < tb ? A.b_tok_id[toff + (int64_t)j * 64] : -1;
                    for (int i = 0; i < ta; ++i)
                        if (s_tid[i] == idb) {
                            ca |= 1u << i;
                            cb |= 1u << j;
                        }
                }
            };

            auto token_set = [&]() -> double {
                if (ta == 0 || tb == 0) return 0.0;
                const int nc = __popc(ca);
                if (nc > 0 && (nc == ta || nc == tb)) return 100.0;
                // lengths of the joined differences and of the joined intersection
                const uint32_t ra = ~ca & (ta >= 32 ? ~0u : ((1u << ta) - 1u)), rb = ~cb & (tb >= 32 ? ~0u : ((1u << tb) - 1u));
                int ab_len = __popc(ra) - 1, ba_len = __popc(rb) - 1, sect_len = nc > 0 ? nc - 1 : 0;
                uint64_t amask[W];
#pragma unroll
                for (int w = 0; w < W; ++w) amask[w] = 0ull;
                const int last_ra = 31 - __clz(ra), last_rb = 31 - __clz(rb);
                for (int i = 0; i < ta; ++i) {
                    const bool rem = (ra >> i) & 1u;
                    ab_len += rem ? s_tlen[i] : 0;
                    sect_len += rem ? 0 : s_tlen[i];
#pragma unroll
                    for (int w = 0; w < W; ++w) amask[w] |= rem ? (s_tmask[i][w] | (i != last_ra ? s_smask[i][w] : 0ull)) : 0ull;
                }
                const int tmax = __builtin_amdgcn_readfirstlane(A.b_tgmax[g]);
                const int64_t toff = A.b_tgoff[g] + lane;
                for (int j = 0; j < tmax; ++j)
                    if (j < tb && ((rb >> j) & 1u)) ba_len += A.b_tok_len[toff + (int64_t)j * 64];
                uint64_t V[W];
                lcs_pass(2, amask, true, rb, last_rb, V);
                const int lcs = zeros_below<W>(V, la2);
                const int sect_sep = sect_len != 0 ? 1 : 0;
                const int sect_ab_len = sect_len + sect_sep + ab_len, sect_ba_len = sect_len + sect_sep + ba_len;
                const double result = norm_distance(ab_len + ba_len - 2 * lcs, sect_ab_len + sect_ba_len);
                if (sect_len == 0) return result;
                const double r_ab = norm_distance(sect_sep + ab_len, sect_len + sect_ab_len);
                const double r_ba = norm_distance(sect_sep + ba_len, sect_len + sect_ba_len);
                return fmax(result, fmax(r_ab, r_ba));
            };
            auto token_sort = [&]() -> double {
                uint64_t V[W];
                lcs_pass(1, all, false, 0u, 0, V);
                return ratio_of(zeros_below<W>(V, la1), la1 + lb1);
            };
            // partial_ratio, left out (0: a lower bound) when it cannot reach the workgroup's best score: a window has
            // at most the LCS of the whole strings -- one cheap pass -- and at least that many characters
            auto partial_pruned = [&](int v, int la, int lb) -> double {
                if (la == 0 || lb == 0) return la == 0 && lb == 0 ? 100.0 : 0.0;
                uint64_t Vv[W];
                lcs_pass(v, all, false, 0u, 0, Vv);
                const int l = zeros_below<W>(Vv, la), lm = min(la, lb);
                const bool want = !(ratio_of(l, lm + l) < cur);
                double p = 0.0;
                if (__any(want)) p = partial(v, want ? la : 0, want ? lb : 0);
                return want ? p : 0.0;
            };
            auto partial_token = [&]() -> double {          // partial_token_ratio
                if (ta == 0 || tb == 0) return 0.0;
                if (ca) return 100.0;
                return fmax(partial_pruned(1, la1, lb1), partial_pruned(2, la2, lb2));
            };

            double score = 0.0;
            if (mode != kPartialRatio && mode != kPartialTokenSortRatio) intersect();
            if (mode == kWRatio) {
                if (la0 != 0 && lb0 != 0) {
                    uint64_t V[W];
                    lcs_pass(0, all, false, 0u, 0, V);
                    double end_ratio = ratio_of(zeros_below<W>(V, la0), la0 + lb0);
                    const int lmax = max(la0, lb0), lmin = min(la0, lb0);
                    // (uniformity: lanes of a group have similar lengths, most groups take one branch as a whole)
                    const bool near = 2 * lmax < 3 * lmin;                  // len_ratio < 1.5
                    // extractOne keeps the maximum, so a COMPONENT of this pair's score that cannot reach `cur` -- a score
                    // some valid choice of this from-string already has -- need not be computed: if the pair wins, it wins
                    // through another component.  Bounds (each through the same floating-point expressions as the value
                    // it bounds, which are monotone):  token scorers <= 100;  a window of either string has LCS <= the
                    // LCS of the whole strings (just computed) and at least that many characters, so partial_ratio <=
                    // ratio_of(lcs, |shorter| + lcs);  the distinct-token form is a subsequence of the sorted-token form.
                    const int lcs0 = zeros_below<W>(V, la0);
                    const double scale = lmax < 8 * lmin ? 0.9 : 0.6;       // len_ratio < 8
                    const bool want_tok = near && !(100.0 * 0.95 < cur);
                    const bool want_ps = !near && !(ratio_of(lcs0, lmin + lcs0) * scale < cur);
                    double tok = 0.0, ps = 0.0, pt = 0.0;
                    if (__any(want_tok)) {
                        const double t1 = token_sort(), t2 = token_set();     // (no tokens on either side: ratio("", "") = 100, as rapidfuzz)
                        tok = fmax(t1, t2);
                    }
                    if (__any(want_ps)) ps = partial(0, want_ps ? la0 : 0, want_ps ? lb0 : 0);
                    if (!near && ta != 0 && tb != 0) {
                        if (ca) pt = 100.0;
                        else if (!(100.0 * 0.95 * scale < cur)) {
                            uint64_t V1[W];
                            lcs_pass(1, all, false, 0u, 0, V1);
                            const int lcs1 = zeros_below<W>(V1, la1);
                            const int c1 = min(lcs1, min(la1, lb1)), c2 = min(lcs1, min(la2, lb2));
                            const bool want1 = !(ratio_of(c1, min(la1, lb1) + c1) * 0.95 * scale < cur);
                            const bool want2 = !(ratio_of(c2, min(la2, lb2) + c2) * 0.95 * scale < cur);
                            double p1 = 0.0, p2 = 0.0;
                            if (__any(want1)) p1 = partial(1, want1 ? la1 : 0, want1 ? lb1 : 0);
                            if (__any(want2)) p2 = partial(2, want2 ? la2 : 0, want2 ? lb2 : 0);
                            pt = fmax(want1 ? p1 : 0.0, want2 ? p2 : 0.0);
                        }
                    }
                    if (near)
                        score = want_tok ? fmax(end_ratio, tok * 0.95) : end_ratio;
                    else {
                        end_ratio = fmax(end_ratio, (want_ps ? ps : 0.0) * scale);
                        score = fmax(end_ratio, pt * 0.95 * scale);
                    }
                }
            }
            else if (mode == kPartialRatio) score = partial_pruned(0, la0, lb0);
            else if (mode == kTokenSetRatio) score = token_set();
            else if (mode == kTokenRatio) score = fmax(token_sort(), token_set());
            else if (mode == kPartialTokenSortRatio) score = partial_pruned(1, la1, lb1);
            else if (mode == kPartialTokenSetRatio)
                score = (ta == 0 || tb == 0) ? 0.0 : (ca ? 100.0 : partial_pruned(2, la2, lb2));
            else score = partial_token();

            if (orig >= 0 && orig != skip && (score > best_score || (score == best_score && orig < best_idx))) {
                best_score = score;
                best_idx = orig;
            }
            if (pruning) {                  // publish the wave's best score to the workgroup
                double wb = fmax(best_score, 0.0);
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) wb = fmax(wb, __shfl_xor(wb, d, 64));
                if (lane == 0) atomicMax(&s_best, (unsigned long long)__double_as_longlong(wb));
            }
        }
        // first best choice: (score desc, original index asc)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double os = __shfl_xor(best_score, d, 64);
            const int oi = __shfl_xor(best_idx, d, 64);
            if (os > best_score || (os == best_score && oi < best_idx)) {
                best_score = os;
                best_idx = oi;
            }
        }
        if (lane == 0) {
            red_s[wave] = best_score;
            red_i[wave] = best_idx;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (red_s[w] > best_score || (red_s[w] == best_score && red_i[w] < best_idx)) {
                    best_score = red_s[w];
                    best_idx = red_i[w];
                }
            if (parts > 1) {
                A.part_score[(int64_t)r * parts + part] = best_score;
                A.part_idx[(int64_t)r * parts + part] = best_idx;
            }
            else {
                A.out_idx[row] = best_idx == INT_MAX ? -1 : best_idx;
                A.out_score[row] = best_idx == INT_MAX ? 0.0 : best_score;
            }
        }
        // clear this from-string's table entries
        for (int v = 0; v < 3; ++v) {
            const int64_t a0 = A.a_off[v][row];
            const int m = (int)(A.a_off[v][row + 1] - a0);
            for (int p = tid; p < m; p += 256) {
                const int sy = A.a_sym[v][a0 + p];
#pragma unroll
                for (int w = 0; w < W; ++w) pm[(sy * 3 + v) * W + w] = 0ull;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k7_merge_parts(FuzzArgs A)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= A.n_rows) return;
    double bs = -1.0;
    int bi = INT_MAX;
    for (int p = 0; p < A.parts; ++p) {
        const double s = A.part_score[(int64_t)r * A.parts + p];
        const int i = A.part_idx[(int64_t)r * A.parts + p];
        if (s > bs || (s == bs && i < bi)) {
            bs = s;
            bi = i;
        }
    }
    const int row = A.rows[r];
    A.out_idx[row] = bi == INT_MAX ? -1 : bi;
    A.out_score[row] = bi == INT_MAX ? 0.0 : bs;
}

namespace {

struct Dev {
    pfz_ctx *ctx;
    std::vector<void *> owned;
    explicit Dev(pfz_ctx *c) : ctx(c) {}
    ~Dev() { for (void *p : owned) pool_free(p); }
    template <typename T> int up(const T *src, size_t n, const T **out)
    {
        void *p = nullptr;
        PFZ_TRY(pool_alloc_raw(ctx, &p, std::max<size_t>(n, 1) * sizeof(T) + 256));
        owned.push_back(p);
        if (n) PFZ_TRY(copy_h2d(ctx, p, src, n * sizeof(T)));
        *out = (const T *)p;
        return PFZ_OK;
    }
    template <typename T> int up(const std::vector<T> &v, const T **out) { return up(v.data(), v.size(), out); }
    template <typename T> int alloc(size_t n, T **out)
    {
        void *p = nullptr;
        PFZ_TRY(pool_alloc_raw(ctx, &p, std::max<size_t>(n, 1) * sizeof(T)));
        owned.push_back(p);
        *out = (T *)p;
        return PFZ_OK;
    }
};

int check_list(const pfz_fuzz_list *L, const char *what, int32_t n_sym)
{
    PFZ_REQUIRE(L && L->n >= 0, "pfz_fuzz_extract_one: bad %s list", what);
    if (L->n == 0) return PFZ_OK;
    for (int v = 0; v < 3; ++v) {
        PFZ_REQUIRE(L->sym[v] && L->off[v], "pfz_fuzz_extract_one: %s list, form %d: NULL array", what, v);
        PFZ_REQUIRE(L->off[v][0] == 0, "pfz_fuzz_extract_one: %s list, form %d: offsets do not start at 0", what, v);
        for (int64_t i = 0; i < L->n; ++i)
            PFZ_REQUIRE(L->off[v][i + 1] >= L->off[v][i], "pfz_fuzz_extract_one: %s list, form %d: offsets decrease at %lld", what,
                        v, (long long)i);
        for (int64_t p = 0; p < L->off[v][L->n]; ++p)
            PFZ_REQUIRE(L->sym[v][p] <= n_sym, "pfz_fuzz_extract_one: %s list, form %d: symbol %d at %lld beyond the alphabet of %d",
                        what, v, (int)L->sym[v][p], (long long)p, n_sym);
    }
    PFZ_REQUIRE(L->tok_off && L->tok_off[0] == 0, "pfz_fuzz_extract_one: %s list: token offsets", what);
    for (int64_t i = 0; i < L->n; ++i) {
        const int64_t nt = L->tok_off[i + 1] - L->tok_off[i];
        PFZ_REQUIRE(nt >= 0, "pfz_fuzz_extract_one: %s list: token offsets decrease at %lld", what, (long long)i);
        if (nt > 32) {
            set_error("pfz_fuzz_extract_one: %s string %lld has %lld distinct tokens; the kernel's token sets hold 32", what,
                      (long long)i, (long long)nt);
            return PFZ_ERR_UNSUPPORTED;
        }
        int64_t joined = nt > 0 ? nt - 1 : 0;
        for (int64_t t = L->tok_off[i]; t < L->tok_off[i + 1]; ++t) joined += L->tok_len[t];
        PFZ_REQUIRE(joined == L->off[2][i + 1] - L->off[2][i],
                    "pfz_fuzz_extract_one: %s string %lld: form 2 is not its distinct tokens joined by single spaces", what,
                    (long long)i);
    }
    return PFZ_OK;
}

}  // namespace

}  // namespace pfz

using namespace pfz;

extern "C" int pfz_fuzz_extract_one(pfz_ctx *ctx, const pfz_fuzz_list *from, const pfz_fuzz_list *to, int32_t n_symbols,
                                    int32_t scorer, const int32_t *skip_idx, int32_t *out_idx, double *out_score)
{
    PFZ_REQUIRE(ctx && from && to && out_idx && out_score, "pfz_fuzz_extract_one: NULL argument");
    PFZ_REQUIRE(scorer >= kWRatio && scorer <= kPartialTokenRatio, "pfz_fuzz_extract_one: unknown scorer %d", scorer);
    PFZ_REQUIRE(n_symbols >= 0 && n_symbols < 65535, "pfz_fuzz_extract_one: alphabet of %d symbols", n_symbols);
    PFZ_TRY(check_list(from, "from", n_symbols));
    PFZ_TRY(check_list(to, "to", n_symbols));
    const int64_t n_from = from->n, n_to = to->n;
    if (n_from == 0) return PFZ_OK;
    if (n_to == 0) {
        for (int64_t i = 0; i < n_from; ++i) {
            out_idx[i] = -1;
            out_score[i] = 0.0;
        }
        return PFZ_OK;
    }
    PFZ_REQUIRE(n_to < INT_MAX - 64 && n_from < INT_MAX, "pfz_fuzz_extract_one: lists of more than 2^31 strings");
    PFZ_HIP(hipSetDevice(ctx->device));

    // word classes of the from-strings
    static const int kWords[3] = {1, 2, 4};
    std::vector<int32_t> cls[3];
    for (int64_t i = 0; i < n_from; ++i) {
        int64_t longest = 0;
        for (int v = 0; v < 3; ++v) longest = std::max(longest, from->off[v][i + 1] - from->off[v][i]);
        if (longest > 256) {
            set_error("pfz_fuzz_extract_one: from-string %lld has %lld characters; the kernel holds 256 (four 64-bit words)",
                      (long long)i, (long long)longest);
            return PFZ_ERR_UNSUPPORTED;
        }
        cls[longest > 128 ? 2 : longest > 64 ? 1 : 0].push_back((int32_t)i);
    }
    for (int c = 0; c < 3; ++c)
        if (!cls[c].empty() && (size_t)(n_symbols + 1) * 3 * kWords[c] * sizeof(uint64_t) > 60 * 1024) {
            set_error("pfz_fuzz_extract_one: an alphabet of %d symbols x 3 forms x %d words does not fit the 60 KiB match table",
                      n_symbols, kWords[c]);
            return PFZ_ERR_UNSUPPORTED;
        }

    // ---- to-side: groups of 64 sorted by the length of form 0, everything [position][lane]
    std::vector<int32_t> order((size_t)n_to);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
        return to->off[0][x + 1] - to->off[0][x] < to->off[0][y + 1] - to->off[0][y];
    });
    const int64_t n_groups = (n_to + 63) / 64;
    std::vector<int32_t> b_orig((size_t)n_groups * 64, -1), b_ntok((size_t)n_groups * 64, 0), b_len[3], b_gmax[3], b_tgmax((size_t)n_groups, 0);
    std::vector<int64_t> b_goff[3], b_tgoff((size_t)n_groups, 0);
    std::vector<uint16_t> b_sym[3];
    std::vector<uint8_t> b_tag;
    for (int v = 0; v < 3; ++v) {
        b_len[v].assign((size_t)n_groups * 64, 0);
        b_gmax[v].assign((size_t)n_groups, 0);
        b_goff[v].assign((size_t)n_groups, 0);
    }
    int64_t tok_total = 0;
    for (int64_t g = 0; g < n_groups; ++g) {
        for (int l = 0; l < 64 && g * 64 + l < n_to; ++l) {
            const int32_t o = order[(size_t)(g * 64 + l)];
            b_orig[(size_t)(g * 64 + l)] = o;
            for (int v = 0; v < 3; ++v) {
                const int32_t len = (int32_t)(to->off[v][o + 1] - to->off[v][o]);
                b_len[v][(size_t)(g * 64 + l)] = len;
                b_gmax[v][(size_t)g] = std::max(b_gmax[v][(size_t)g], len);
            }
            const int32_t nt = (int32_t)(to->tok_off[o + 1] - to->tok_off[o]);
            b_ntok[(size_t)(g * 64 + l)] = nt;
            b_tgmax[(size_t)g] = std::max(b_tgmax[(size_t)g], nt);
        }
        b_tgoff[(size_t)g] = tok_total;
        tok_total += (int64_t)b_tgmax[(size_t)g] * 64;
    }
    std::vector<int32_t> b_tok_id((size_t)tok_total + 64, -1), b_tok_len((size_t)tok_total + 64, 0);
    for (int v = 0; v < 3; ++v) {
        int64_t total = 0;
        for (int64_t g = 0; g < n_groups; ++g) {
            b_goff[v][(size_t)g] = total;
            total += (int64_t)b_gmax[v][(size_t)g] * 64;
        }
        b_sym[v].assign((size_t)total + 64, 0);
        if (v == 2) b_tag.assign((size_t)total + 64, 0);
    }
    for (int64_t g = 0; g < n_groups; ++g)
        for (int l = 0; l < 64 && g * 64 + l < n_to; ++l) {
            const int32_t o = order[(size_t)(g * 64 + l)];
            for (int v = 0; v < 3; ++v) {
                const int64_t s0 = to->off[v][o];
                const int32_t len = b_len[v][(size_t)(g * 64 + l)];
                for (int32_t p = 0; p < len; ++p) b_sym[v][(size_t)(b_goff[v][(size_t)g] + (int64_t)p * 64 + l)] = to->sym[v][s0 + p];
            }
            const int64_t t0 = to->tok_off[o];
            const int32_t nt = b_ntok[(size_t)(g * 64 + l)];
            int32_t pos = 0;
            for (int32_t j = 0; j < nt; ++j) {
                const int32_t len = to->tok_len[t0 + j];
                b_tok_id[(size_t)(b_tgoff[(size_t)g] + (int64_t)j * 64 + l)] = to->tok_id[t0 + j];
                b_tok_len[(size_t)(b_tgoff[(size_t)g] + (int64_t)j * 64 + l)] = len;
                for (int32_t p = 0; p < len; ++p) b_tag[(size_t)(b_goff[2][(size_t)g] + (int64_t)(pos + p) * 64 + l)] = (uint8_t)j;
                if (j + 1 < nt) b_tag[(size_t)(b_goff[2][(size_t)g] + (int64_t)(pos + len) * 64 + l)] = (uint8_t)(j | 0x80);
                pos += len + 1;
            }
        }

    Dev dev(ctx);
    FuzzArgs A{};
    for (int v = 0; v < 3; ++v) {
        PFZ_TRY(dev.up(from->sym[v], (size_t)from->off[v][n_from], &A.a_sym[v]));
        PFZ_TRY(dev.up(from->off[v], (size_t)n_from + 1, &A.a_off[v]));
        PFZ_TRY(dev.up(b_sym[v], &A.b_sym[v]));
        PFZ_TRY(dev.up(b_goff[v], &A.b_goff[v]));
        PFZ_TRY(dev.up(b_gmax[v], &A.b_gmax[v]));
        PFZ_TRY(dev.up(b_len[v], &A.b_len[v]));
    }
    PFZ_TRY(dev.up(from->tok_id, (size_t)from->tok_off[n_from], &A.a_tok_id));
    PFZ_TRY(dev.up(from->tok_len, (size_t)from->tok_off[n_from], &A.a_tok_len));
    PFZ_TRY(dev.up(from->tok_off, (size_t)n_from + 1, &A.a_tok_off));
    PFZ_TRY(dev.up(b_tag, &A.b_tag));
    PFZ_TRY(dev.up(b_orig, &A.b_orig));
    PFZ_TRY(dev.up(b_ntok, &A.b_ntok));
    PFZ_TRY(dev.up(b_tgoff, &A.b_tgoff));
    PFZ_TRY(dev.up(b_tgmax, &A.b_tgmax));
    PFZ_TRY(dev.up(b_tok_id, &A.b_tok_id));
    PFZ_TRY(dev.up(b_tok_len, &A.b_tok_len));
    if (skip_idx) PFZ_TRY(dev.up(skip_idx, (size_t)n_from, &A.skip_idx));
    PFZ_TRY(dev.alloc((size_t)n_from, &A.out_idx));
    PFZ_TRY(dev.alloc((size_t)n_from, &A.out_score));
    A.n_groups = (int32_t)n_groups;
    A.n_sym1 = n_symbols + 1;
    A.mode = scorer;

    const int64_t max_grid = (int64_t)ctx->prop.multiProcessorCount * 4;
    for (int c = 0; c < 3; ++c) {
        if (cls[c].empty()) continue;
        PFZ_TRY(dev.up(cls[c], &A.rows));
        A.n_rows = (int32_t)cls[c].size();
        // few from-strings: split every string's to-groups over `parts` workgroups (as K4 does)
        const int64_t want = (2 * max_grid + A.n_rows - 1) / A.n_rows, cap = std::max<int64_t>(1, n_groups / 4);
        A.parts = (int32_t)std::max<int64_t>(1, std::min(want, cap));
        if (const char *e = getenv("PFZ_K7_PARTS")) A.parts = std::max(1, atoi(e));
        if (A.parts > 1) {
            PFZ_TRY(dev.alloc((size_t)A.n_rows * (size_t)A.parts, &A.part_score));
            PFZ_TRY(dev.alloc((size_t)A.n_rows * (size_t)A.parts, &A.part_idx));
        }
        const unsigned grid = (unsigned)std::min<int64_t>((int64_t)A.n_rows * A.parts, max_grid);
        const size_t lds = (size_t)A.n_sym1 * 3 * (size_t)kWords[c] * sizeof(uint64_t);
        {
            ProfScope ps(ctx, "k7_fuzz");
            if (c == 0) hipLaunchKernelGGL(k7_fuzz_kernel<1>, dim3(grid), dim3(256), lds, ctx->stream, A);
            else if (c == 1) hipLaunchKernelGGL(k7_fuzz_kernel<2>, dim3(grid), dim3(256), lds, ctx->stream, A);
            else hipLaunchKernelGGL(k7_fuzz_kernel<4>, dim3(grid), dim3(256), lds, ctx->stream, A);
            PFZ_HIP(hipGetLastError());
            if (A.parts > 1) {
                hipLaunchKernelGGL(k7_merge_parts, dim3((unsigned)((A.n_rows + 255) / 256)), dim3(256), 0, ctx->stream, A);
                PFZ_HIP(hipGetLastError());
            }
        }
    }
    PFZ_TRY(copy_d2h(ctx, out_idx, A.out_idx, (size_t)n_from * sizeof(int32_t)));
    PFZ_TRY(copy_d2h(ctx, out_score, A.out_score, (size_t)n_from * sizeof(double)));
    PFZ_HIP(hipStreamSynchronize(ctx->stream));
    return PFZ_OK;
}
