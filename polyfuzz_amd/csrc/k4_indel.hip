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
//     points of both lists, so any Unicode input works), built with ds_or and
//     cleared by re-visiting the from-string's own characters.
//   lane = one to-string.  To-strings are sorted by length and stored in groups
//     of 64 as [t/4][lane] dwords of 4 packed symbols (2 for >255 symbols), so a
//     wave reads 256 contiguous bytes per step and lanes of a group finish
//     together; padding symbol 0 has an empty PM entry and is a no-op.
//   from-strings are bucketed by length into word classes (32-bit word for <= 32
//     characters, 1..16 64-bit words beyond) so V stays in registers.
//   arg-max: per lane (score desc, original index asc) in float64 with the exact
//     reference formula, then wave shuffles + one LDS step per from-string.
// Roofline: integer VALU + LDS lookups; HBM traffic is the to-strings once per
// from-string out of L2 (0.3 MB) -- not HBM-bound.
#include "pfz_internal.h"

#include <algorithm>
#include <limits.h>
#include <string.h>

namespace pfz {

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
        V[w] = sum | (s ^ u);
    }
}

__device__ inline void lds_or(uint32_t *p, uint32_t v) { atomicOr(p, v); }
__device__ inline void lds_or(uint64_t *p, uint64_t v) { atomicOr((unsigned long long *)p, (unsigned long long)v); }
__device__ inline int popc_word(uint32_t v) { return __popc(v); }
__device__ inline int popc_word(uint64_t v) { return __popcll(v); }

struct IndelArgs {
    const uint16_t *a_ids;     // from-strings as symbol ranks
    const int64_t *a_off;      // [n_from + 1]
    const int32_t *rows;       // from-rows of this word class
    int32_t n_rows;
    const uint32_t *b_packed;  // to-strings, groups of 64, [t/PER][lane]
    const int64_t *g_off;      // [n_groups] dword offset of each group
    const int32_t *g_steps;    // [n_groups] dwords per lane
    const int32_t *b_len;      // [n_groups*64]
    const int32_t *b_orig;     // [n_groups*64] original to-index, -1 = padding lane
    int32_t n_groups;
    const int32_t *skip_idx;   // [n_from] or NULL
    int32_t n_sym1;            // alphabet size + 1 (symbol 0 = padding)
    int64_t from_begin;
    int64_t n_to;
    int32_t *out_idx;          // [from_end - from_begin]
    double *out_score;
    double *matrix;            // optional [(from_end-from_begin) * n_to]
};

template <typename WORD, int W, int IDB>
__global__ __launch_bounds__(256) void k4_indel_kernel(IndelArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    WORD *pm = (WORD *)smem_raw;
    __shared__ double red_s[4];
    __shared__ int red_i[4];
    constexpr int WB = sizeof(WORD) * 8;
    constexpr int PER = 32 / IDB;  // symbols per dword
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int p = tid; p < A.n_sym1 * W; p += 256) pm[p] = 0;
    __syncthreads();

    for (int r = blockIdx.x; r < A.n_rows; r += gridDim.x) {
        const int row = A.rows[r];
        const uint16_t *a = A.a_ids + A.a_off[row];
        const int m = (int)(A.a_off[row + 1] - A.a_off[row]);
        for (int p = tid; p < m; p += 256) lds_or(&pm[(int)a[p] * W + p / WB], (WORD)1 << (p % WB));
        __syncthreads();

        const int skip = A.skip_idx ? A.skip_idx[row] : -1;
        double best = -1.0;
        int besti = INT_MAX;
        for (int g = wave; g < A.n_groups; g += 4) {
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
                const int64_t maximum = (int64_t)m + lb;
                const int64_t dist = maximum - 2 * (int64_t)lcs;
                // rapidfuzz: norm_dist = dist / maximum (0 when both empty); ratio = (1 - norm_dist) * 100
                const double norm_dist = maximum != 0 ? (double)dist / (double)maximum : 0.0;
                const double score = (1.0 - norm_dist) * 100.0;
                if (A.matrix) A.matrix[((int64_t)row - A.from_begin) * A.n_to + orig] = orig == skip ? -1.0 : score;
                if (orig != skip && (score > best || (score == best && orig < besti))) {
                    best = score;
                    besti = orig;
                }
            }
        }
        // first maximum: (score desc, original index asc)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double os = __shfl_xor(best, d, 64);
            const int oi = __shfl_xor(besti, d, 64);
            if (os > best || (os == best && oi < besti)) {
                best = os;
                besti = oi;
            }
        }
        if (lane == 0) {
            red_s[wave] = best;
            red_i[wave] = besti;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (red_s[w] > best || (red_s[w] == best && red_i[w] < besti)) {
                    best = red_s[w];
                    besti = red_i[w];
                }
            const int64_t o = (int64_t)row - A.from_begin;
            A.out_idx[o] = besti == INT_MAX ? -1 : besti;
            A.out_score[o] = besti == INT_MAX ? 0.0 : best;
        }
        // clear the PM entries of this from-string
        for (int p = tid; p < m; p += 256) {
#pragma unroll
            for (int w = 0; w < W; ++w) pm[(int)a[p] * W + w] = 0;
        }
        __syncthreads();
    }
}

// ---- host side ---------------------------------------------------------------

static inline uint32_t unit_at(const pfz_strings *s, int64_t p)
{
    if (s->char_width == 1) return s->h_chars[(size_t)p];
    uint32_t v;
    memcpy(&v, &s->h_chars[(size_t)p * 4], 4);
    return v;
}

struct DevBuf {
    pfz_ctx *ctx = nullptr;
    void *p = nullptr;
    ~DevBuf() { if (p) pool_free(p); }
    int alloc(size_t bytes) { return pool_alloc(ctx, &p, bytes > 0 ? bytes : 16); }
    template <typename T> int upload(const std::vector<T> &v, hipStream_t st)
    {
        PFZ_TRY(alloc(v.size() * sizeof(T)));
        (void)st;
        if (!v.empty()) PFZ_TRY(copy_h2d(ctx, p, v.data(), v.size() * sizeof(T)));   // ctx->stream; v may be freed on return
        return PFZ_OK;
    }
};

template <typename WORD, int W>
static int launch_class(pfz_ctx *ctx, const IndelArgs &A, int idb, unsigned grid)
{
    const size_t lds = (size_t)A.n_sym1 * W * sizeof(WORD);
    if (lds > 60 * 1024) {
        set_error("pfz_indel: %d alphabet symbols x %d words need %zu bytes of LDS for the match table (limit 60 KiB)",
                  A.n_sym1 - 1, W, lds);
        return PFZ_ERR_UNSUPPORTED;
    }
    ProfScope ps(ctx, "k4_indel");
    if (idb == 8)
        hipLaunchKernelGGL((k4_indel_kernel<WORD, W, 8>), dim3(grid), dim3(256), lds, ctx->stream, A);
    else
        hipLaunchKernelGGL((k4_indel_kernel<WORD, W, 16>), dim3(grid), dim3(256), lds, ctx->stream, A);
    PFZ_HIP(hipGetLastError());
    return PFZ_OK;
}

static int indel_run(pfz_ctx *ctx, const pfz_strings *F, const pfz_strings *T, const int32_t *skip_idx, int64_t begin,
                     int64_t end, int32_t *out_idx, double *out_score, double *out_matrix)
{
    PFZ_REQUIRE(ctx && F && T, "pfz_indel: NULL argument");
    PFZ_REQUIRE(begin >= 0 && begin <= end && end <= F->n, "pfz_indel: row range [%lld,%lld) outside [0,%lld)",
                (long long)begin, (long long)end, (long long)F->n);
    const int64_t n_rows = end - begin;
    if (n_rows == 0) return PFZ_OK;
    PFZ_HIP(hipSetDevice(ctx->device));
    if (T->n >= INT_MAX - 64 || F->n >= INT_MAX) {
        set_error("pfz_indel: list too long");
        return PFZ_ERR_UNSUPPORTED;
    }

    // alphabet: rank (1..S) of every distinct code point of both lists
    std::vector<uint32_t> cps;
    {
        // (typed loops over the host mirror and a scan that stops at the largest code point seen: the generic
        // per-unit accessor + a walk over all 1.1 M code points cost 2-15 ms per call)
        const bool any_wide = F->char_width != 1 || T->char_width != 1;
        std::vector<uint8_t> seen(any_wide ? 0x110000 / 8 + 1 : 32, 0);
        uint32_t max_cp = 0;
        for (const pfz_strings *s : {F, T}) {
            if (s->char_width == 1) {
                const uint8_t *u = s->h_chars.data();
                for (int64_t p = 0; p < s->n_units; ++p) seen[u[p] >> 3] |= (uint8_t)(1u << (u[p] & 7));
                if (s->n_units > 0 && max_cp < 255) max_cp = 255;
            } else {
                const uint32_t *u = (const uint32_t *)s->h_chars.data();
                for (int64_t p = 0; p < s->n_units; ++p) {
                    const uint32_t c = u[p];
                    if (c < 0x110000u) {
                        seen[c >> 3] |= (uint8_t)(1u << (c & 7));
                        max_cp = c > max_cp ? c : max_cp;
                    }
                }
            }
        }
        for (uint32_t c = 0; c <= max_cp; ++c)
            if (seen[c >> 3] & (1u << (c & 7))) cps.push_back(c);
    }
    const int S = (int)cps.size();
    if (S > 65535) {
        set_error("pfz_indel: %d distinct code points exceed the 16-bit symbol space", S);
        return PFZ_ERR_UNSUPPORTED;
    }
    const int idb = S <= 255 ? 8 : 16;
    const int per = 32 / idb;
    auto sym = [&](uint32_t c) -> uint16_t {
        return (uint16_t)(std::lower_bound(cps.begin(), cps.end(), c) - cps.begin() + 1);
    };
    std::vector<uint16_t> lut;
    if (!cps.empty() && cps.back() < 65536) {
        lut.assign((size_t)cps.back() + 1, 0);
        for (size_t r = 0; r < cps.size(); ++r) lut[cps[r]] = (uint16_t)(r + 1);
    }
    auto sym_fast = [&](uint32_t c) -> uint16_t { return !lut.empty() ? lut[c] : sym(c); };

    // from side: symbol arrays + word classes by length
    std::vector<uint16_t> a_ids((size_t)F->n_units);
    if (F->char_width == 1 && !lut.empty()) {
        const uint8_t *u = F->h_chars.data();
        for (int64_t p = 0; p < F->n_units; ++p) a_ids[(size_t)p] = lut[u[p]];
    } else {
        for (int64_t p = 0; p < F->n_units; ++p) a_ids[(size_t)p] = sym_fast(unit_at(F, p));
    }
    static const int kClassMax[7] = {32, 64, 128, 256, 512, 1024, INT_MAX};
    std::vector<int32_t> cls[7];
    for (int64_t i = begin; i < end; ++i) {
        const int64_t m = F->h_off[(size_t)i + 1] - F->h_off[(size_t)i];
        int c = 0;
        while (m > kClassMax[c]) ++c;
        cls[c].push_back((int32_t)i);
    }
    if (!cls[6].empty()) {
        set_error("pfz_indel: from-string %d has more than 1024 characters (bit-parallel word classes cover <= 1024)",
                  cls[6][0]);
        return PFZ_ERR_UNSUPPORTED;
    }

    // to side: sort by length, groups of 64, [step][lane] dwords of packed symbols
    const int64_t n_to = T->n;
    std::vector<int32_t> order((size_t)n_to);
    for (int64_t j = 0; j < n_to; ++j) order[(size_t)j] = (int32_t)j;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
        return (T->h_off[(size_t)x + 1] - T->h_off[(size_t)x]) < (T->h_off[(size_t)y + 1] - T->h_off[(size_t)y]);
    });
    const int64_t n_groups = (n_to + 63) / 64;
    std::vector<int64_t> g_off((size_t)n_groups);
    std::vector<int32_t> g_steps((size_t)n_groups), b_len((size_t)n_groups * 64, 0), b_orig((size_t)n_groups * 64, -1);
    int64_t total_dw = 0;
    for (int64_t g = 0; g < n_groups; ++g) {
        int64_t mx = 0;
        for (int l = 0; l < 64 && g * 64 + l < n_to; ++l) {
            const int32_t j = order[(size_t)(g * 64 + l)];
            const int64_t len = T->h_off[(size_t)j + 1] - T->h_off[(size_t)j];
            b_len[(size_t)(g * 64 + l)] = (int32_t)len;
            b_orig[(size_t)(g * 64 + l)] = j;
            mx = std::max(mx, len);
        }
        g_off[(size_t)g] = total_dw;
        g_steps[(size_t)g] = (int32_t)((mx + per - 1) / per);
        total_dw += (int64_t)g_steps[(size_t)g] * 64;
    }
    std::vector<uint32_t> packed((size_t)total_dw, 0u);
    for (int64_t g = 0; g < n_groups; ++g)
        for (int l = 0; l < 64 && g * 64 + l < n_to; ++l) {
            const int32_t j = order[(size_t)(g * 64 + l)];
            const int64_t b0 = T->h_off[(size_t)j], len = T->h_off[(size_t)j + 1] - b0;
            uint32_t *dst = packed.data() + g_off[(size_t)g] + l;
            if (T->char_width == 1 && !lut.empty()) {
                const uint8_t *u = T->h_chars.data() + b0;
                for (int64_t t = 0; t < len; ++t) dst[(t / per) * 64] |= (uint32_t)lut[u[t]] << ((t % per) * idb);
            } else {
                for (int64_t t = 0; t < len; ++t)
                    dst[(t / per) * 64] |= (uint32_t)sym_fast(unit_at(T, b0 + t)) << ((t % per) * idb);
            }
        }

    DevBuf d_a, d_aoff, d_packed, d_goff, d_gsteps, d_blen, d_borig, d_skip, d_oidx, d_oscore, d_matrix, d_rows[6];
    for (DevBuf *b : {&d_a, &d_aoff, &d_packed, &d_goff, &d_gsteps, &d_blen, &d_borig, &d_skip, &d_oidx, &d_oscore,
                      &d_matrix, &d_rows[0], &d_rows[1], &d_rows[2], &d_rows[3], &d_rows[4], &d_rows[5]})
        b->ctx = ctx;
    PFZ_TRY(d_a.upload(a_ids, ctx->stream));
    PFZ_TRY(d_aoff.upload(F->h_off, ctx->stream));
    PFZ_TRY(d_packed.upload(packed, ctx->stream));
    PFZ_TRY(d_goff.upload(g_off, ctx->stream));
    PFZ_TRY(d_gsteps.upload(g_steps, ctx->stream));
    PFZ_TRY(d_blen.upload(b_len, ctx->stream));
    PFZ_TRY(d_borig.upload(b_orig, ctx->stream));
    if (skip_idx) {
        std::vector<int32_t> sk(skip_idx, skip_idx + F->n);
        PFZ_TRY(d_skip.upload(sk, ctx->stream));
    }
    PFZ_TRY(d_oidx.alloc((size_t)n_rows * sizeof(int32_t)));
    PFZ_TRY(d_oscore.alloc((size_t)n_rows * sizeof(double)));
    if (out_matrix) PFZ_TRY(d_matrix.alloc((size_t)n_rows * (size_t)n_to * sizeof(double)));

    IndelArgs A;
    A.a_ids = (const uint16_t *)d_a.p;
    A.a_off = (const int64_t *)d_aoff.p;
    A.b_packed = (const uint32_t *)d_packed.p;
    A.g_off = (const int64_t *)d_goff.p;
    A.g_steps = (const int32_t *)d_gsteps.p;
    A.b_len = (const int32_t *)d_blen.p;
    A.b_orig = (const int32_t *)d_borig.p;
    A.n_groups = (int32_t)n_groups;
    A.skip_idx = skip_idx ? (const int32_t *)d_skip.p : nullptr;
    A.n_sym1 = S + 1;
    A.from_begin = begin;
    A.n_to = n_to;
    A.out_idx = (int32_t *)d_oidx.p;
    A.out_score = (double *)d_oscore.p;
    A.matrix = out_matrix ? (double *)d_matrix.p : nullptr;
    const int64_t max_grid = (int64_t)ctx->prop.multiProcessorCount * 8;
    for (int c = 0; c < 6; ++c) {
        if (cls[c].empty()) continue;
        PFZ_TRY(d_rows[c].upload(cls[c], ctx->stream));
        A.rows = (const int32_t *)d_rows[c].p;
        A.n_rows = (int32_t)cls[c].size();
        const unsigned grid = (unsigned)std::min<int64_t>(A.n_rows, max_grid);
        switch (c) {
        case 0: PFZ_TRY((launch_class<uint32_t, 1>(ctx, A, idb, grid))); break;
        case 1: PFZ_TRY((launch_class<uint64_t, 1>(ctx, A, idb, grid))); break;
        case 2: PFZ_TRY((launch_class<uint64_t, 2>(ctx, A, idb, grid))); break;
        case 3: PFZ_TRY((launch_class<uint64_t, 4>(ctx, A, idb, grid))); break;
        case 4: PFZ_TRY((launch_class<uint64_t, 8>(ctx, A, idb, grid))); break;
        default: PFZ_TRY((launch_class<uint64_t, 16>(ctx, A, idb, grid))); break;
        }
    }
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

int pfz_indel_matrix_host(pfz_ctx *ctx, const pfz_strings *from_strings, const pfz_strings *to_strings,
                          int64_t from_begin, int64_t from_end, double *out_matrix)
{
    PFZ_REQUIRE(out_matrix, "pfz_indel_matrix_host: NULL output");
    if (to_strings && to_strings->n == 0) return PFZ_OK;
    return indel_run(ctx, from_strings, to_strings, nullptr, from_begin, from_end, nullptr, nullptr, out_matrix);
}

}  // extern "C"
