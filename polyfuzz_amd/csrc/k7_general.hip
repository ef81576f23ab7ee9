// K7, the general case: from-strings beyond 256 characters or 32 distinct tokens, and -- for every from-string -- the
// to-strings with more than 32 distinct tokens, which the register / LDS kernel of k7_fuzz.hip leaves out.  The reference
// accepts any string (polyfuzz/models/_rapidfuzz.py:106-108), so the engine does too: here with NO limit on lengths or
// token counts, and no cleverness -- one (from, to) pair per lane, every Indel similarity by the plain O(|a||b|) LCS
// dynamic programme with its row in global scratch, every window of partial_ratio scored on its own, token sets as flag
// bytes in scratch.  It is the per-pair statement of the scorers (rapidfuzz 3.x semantics, as oracle/fuzz_scorers.c)
// over the device-resident forms and plan; slow -- everything is a global-memory access -- and only ever sees the rare
// strings the fast kernel cannot hold.   PARITY UNPINNED (rapidfuzz is not installable).
#include "pfz_internal.h"

#include <algorithm>
#include <climits>

#undef PFZ_HD
#define PFZ_HD __device__ inline
#define PFZ_LDS_U16 __attribute__((address_space(3))) uint16_t
#define PFZ_LDS_U8 __attribute__((address_space(3))) uint8_t
#include "k7_core.h"
#include "k7_args.h"

namespace pfz {

__device__ inline uint32_t g_load_unit(const void *p, int width, int64_t i)
{
    return width == 1 ? (uint32_t)((const uint8_t *)p)[i] : ((const uint32_t *)p)[i];
}


// a string the DP walks: a form of the from-string (code units mapped through the alphabet) or of the lane's to-string
// (symbol ranks, its record of the plan), optionally only the tokens whose flag byte is zero, joined by single spaces
struct Seq {
    // from-form: chars != NULL; to-form: sym != NULL
    const void *chars;
    int width;
    int64_t base;
    const uint16_t *lut;
    uint32_t lut_len;
    const uint16_t *sym;
    int len;
    __device__ int at(int p) const
    {
        if (sym) return (int)sym[p];
        const uint32_t c = g_load_unit(chars, width, base + p);
        return c < lut_len ? (int)lut[c] : -1 - (int)c;      // a from-character outside the to-alphabet matches nothing of the to-side
    }
};

// LCS(x[x0 : x0 + xl], y[y0 : y0 + yl]) by the plain DP; row: yl + 1 ints at row[q * stride]
__device__ int lcs_dp(const Seq &x, int x0, int xl, const Seq &y, int y0, int yl, int32_t *row, int64_t stride)
{
    if (xl == 0 || yl == 0) return 0;
    for (int q = 0; q <= yl; ++q) row[(int64_t)q * stride] = 0;
    for (int i = 0; i < xl; ++i) {
        const int cx = x.at(x0 + i);
        int diag = 0, left = 0;
        for (int q = 0; q < yl; ++q) {
            const int up = row[(int64_t)(q + 1) * stride];
            const int v = cx == y.at(y0 + q) ? diag + 1 : (up > left ? up : left);
            row[(int64_t)(q + 1) * stride] = v;
            diag = up;
            left = v;
        }
    }
    return row[(int64_t)yl * stride];
}

__device__ double g_ratio(const Seq &x, int x0, int xl, const Seq &y, int y0, int yl, int32_t *row, int64_t stride)
{
    return fz_ratio_of(lcs_dp(x, x0, xl, y, y0, yl, row, stride), xl + yl);
}

// len(s1) <= len(s2): prefixes of s2 shorter than s1, its windows of length len(s1), its suffixes shorter than s1
__device__ double g_partial_impl(const Seq &s1, const Seq &s2, int32_t *row, int64_t stride)
{
    double best = 0.0;
    // (the DP row runs over the second argument of lcs_dp: keep it the window, which is never longer than s1)
    for (int i = 1; i < s1.len; ++i) best = fz_fmax(best, g_ratio(s1, 0, s1.len, s2, 0, i, row, stride));
    for (int i = 0; i < s2.len - s1.len; ++i) best = fz_fmax(best, g_ratio(s1, 0, s1.len, s2, i, s1.len, row, stride));
    for (int i = s2.len - s1.len; i < s2.len; ++i) best = fz_fmax(best, g_ratio(s1, 0, s1.len, s2, i, s2.len - i, row, stride));
    return best;
}

__device__ double g_partial_ratio(const Seq &a, const Seq &b, int32_t *row, int64_t stride)
{
    if (a.len == 0 || b.len == 0) return a.len == 0 && b.len == 0 ? 100.0 : 0.0;
    const Seq &shorter = a.len <= b.len ? a : b, &longer = a.len <= b.len ? b : a;
    double res = g_partial_impl(shorter, longer, row, stride);
    if (res != 100.0 && a.len == b.len) res = fz_fmax(res, g_partial_impl(longer, shorter, row, stride));
    return res;
}

// The joined token differences are walked through a position map built once per pair: dpos[k] = position in form 2 of
// the k-th character of " ".join(tokens not common) (the form is the tokens joined by single spaces, so a difference is
// a sub-sequence of its positions: the remaining tokens and the spaces between consecutive remaining ones)
struct Diff {
    Seq form;            // form 2
    const int32_t *pos;  // [k * stride]
    int64_t stride;
    int len;
    __device__ int at(int k) const { return form.at(pos[(int64_t)k * stride]); }
};

__device__ int lcs_dp_diff(const Diff &x, int x0, int xl, const Diff &y, int y0, int yl, int32_t *row, int64_t stride)
{
    if (xl == 0 || yl == 0) return 0;
    for (int q = 0; q <= yl; ++q) row[(int64_t)q * stride] = 0;
    for (int i = 0; i < xl; ++i) {
        const int cx = x.at(x0 + i);
        int diag = 0, left = 0;
        for (int q = 0; q < yl; ++q) {
            const int up = row[(int64_t)(q + 1) * stride];
            const int v = cx == y.at(y0 + q) ? diag + 1 : (up > left ? up : left);
            row[(int64_t)(q + 1) * stride] = v;
            diag = up;
            left = v;
        }
    }
    return row[(int64_t)yl * stride];
}

__device__ double g_partial_impl_diff(const Diff &s1, const Diff &s2, int32_t *row, int64_t stride)
{
    double best = 0.0;
    for (int i = 1; i < s1.len; ++i) best = fz_fmax(best, fz_ratio_of(lcs_dp_diff(s1, 0, s1.len, s2, 0, i, row, stride), s1.len + i));
    for (int i = 0; i < s2.len - s1.len; ++i)
        best = fz_fmax(best, fz_ratio_of(lcs_dp_diff(s1, 0, s1.len, s2, i, s1.len, row, stride), 2 * s1.len));
    for (int i = s2.len - s1.len; i < s2.len; ++i)
        best = fz_fmax(best, fz_ratio_of(lcs_dp_diff(s1, 0, s1.len, s2, i, s2.len - i, row, stride), s1.len + s2.len - i));
    return best;
}

__device__ double g_partial_ratio_diff(const Diff &a, const Diff &b, int32_t *row, int64_t stride)
{
    if (a.len == 0 || b.len == 0) return a.len == 0 && b.len == 0 ? 100.0 : 0.0;
    const Diff &shorter = a.len <= b.len ? a : b, &longer = a.len <= b.len ? b : a;
    double res = g_partial_impl_diff(shorter, longer, row, stride);
    if (res != 100.0 && a.len == b.len) res = fz_fmax(res, g_partial_impl_diff(longer, shorter, row, stride));
    return res;
}

struct GeneralScratch {
    int32_t *row;        // [(max_len + 2)][256] per workgroup: DP row of the lane
    int32_t *dpos_a;     // [max_from_len + 1][256]: positions of the from-side token difference
    int32_t *dpos_b;     // [max_to_len + 1][256]
    int32_t max_row, max_a, max_b;
};

__global__ __launch_bounds__(256) void k7_general_kernel(FuzzArgs A, GeneralScratch S, const int32_t *__restrict__ a_tok_pos)
{
    __shared__ double red_s[4];
    __shared__ int red_i[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mode = A.mode;
    int32_t *row = S.row + (int64_t)blockIdx.x * (S.max_row + 2) * 256 + tid;
    int32_t *dpa = S.dpos_a + (int64_t)blockIdx.x * (S.max_a + 1) * 256 + tid;
    int32_t *dpb = S.dpos_b + (int64_t)blockIdx.x * (S.max_b + 1) * 256 + tid;
    const int64_t stride = 256;
    const int64_t n_slots = A.n_big > 0 ? A.n_big : (int64_t)A.n_groups * 64;
    for (int r = blockIdx.x; r < A.n_rows; r += gridDim.x) {
        const int frow = A.rows[r];
        const int64_t a0 = A.a_off[frow];
        const int la[3] = {(int)(A.a_off[frow + 1] - a0), A.a_len1[frow], A.a_len2[frow]};
        const int ta = A.a_ntok[frow];
        const int64_t atb = (a0 >> 1) + frow;
        const int skip = A.skip_idx ? A.skip_idx[frow] : -1;
        Seq fa[3];
        for (int v = 0; v < 3; ++v) fa[v] = Seq{A.a_form[v], A.a_width, a0, A.lut, A.lut_len, nullptr, la[v]};
        double best_score = -1.0;
        int best_idx = INT_MAX;
        for (int64_t k = tid; k < n_slots; k += 256) {
            const int slot = A.n_big > 0 ? A.big_slots[k] : (int)k;
            const int4 m = A.b_meta[slot];
            const int orig = A.b_meta2[slot].w;
            if (orig < 0 || choice_left_out(orig, skip, A.skip_up_to)) continue;
            const int4 rec = A.b_meta3[slot];
            const int lb[3] = {m.x, m.y, m.z}, tb = m.w;
            Seq fb[3];
            for (int v = 0; v < 3; ++v) fb[v] = Seq{nullptr, 0, 0, nullptr, 0, A.b_sym + rec.x + (int64_t)v * rec.w, lb[v]};
            const int32_t *b_id = A.b_tok_id + rec.z, *b_len = A.b_tok_len + rec.z;

            // the joined differences of the distinct-token sets (and what token_set_ratio needs of the intersection)
            int nc = 0, sect_len = 0;
            Diff da{fa[2], dpa, stride, 0}, db{fb[2], dpb, stride, 0};
            auto differences = [&]() {
                nc = sect_len = 0;
                int na = 0, first = 1;
                for (int i = 0; i < ta; ++i) {
                    const int id = A.a_tok_id[atb + i], len = A.a_tok_len[atb + i], pos = a_tok_pos[atb + i];
                    bool common = false;
                    for (int j = 0; j < tb && !common; ++j) common = b_id[j] == id;
                    if (common) {
                        sect_len += len + (nc ? 1 : 0);
                        ++nc;
                        continue;
                    }
                    if (!first) dpa[(int64_t)(na++) * stride] = pos - 1;          // the space before this token
                    first = 0;
                    for (int q = 0; q < len; ++q) dpa[(int64_t)(na++) * stride] = pos + q;
                }
                da.len = na;
                int nb = 0, posb = 0;
                first = 1;
                for (int j = 0; j < tb; ++j) {
                    const int id = b_id[j], len = b_len[j];
                    bool common = false;
                    for (int i = 0; i < ta && !common; ++i) common = A.a_tok_id[atb + i] == id;
                    if (!common) {
                        if (!first) dpb[(int64_t)(nb++) * stride] = posb - 1;
                        first = 0;
                        for (int q = 0; q < len; ++q) dpb[(int64_t)(nb++) * stride] = posb + q;
                    }
                    posb += len + 1;
                }
                db.len = nb;
            };
            auto token_set = [&]() -> double {
                if (ta == 0 || tb == 0) return 0.0;
                differences();
                if (nc > 0 && (da.len == 0 || db.len == 0)) return 100.0;
                const int ab_len = da.len, ba_len = db.len, sep = sect_len != 0 ? 1 : 0;
                const int sect_ab_len = sect_len + sep + ab_len, sect_ba_len = sect_len + sep + ba_len;
                const int lcs = lcs_dp_diff(da, 0, ab_len, db, 0, ba_len, row, stride);
                const double result = fz_norm_distance(ab_len + ba_len - 2 * lcs, sect_ab_len + sect_ba_len);
                if (!sect_len) return result;
                return fz_fmax(result, fz_fmax(fz_norm_distance(sep + ab_len, sect_len + sect_ab_len),
                                               fz_norm_distance(sep + ba_len, sect_len + sect_ba_len)));
            };
            auto token_sort = [&]() { return g_ratio(fa[1], 0, la[1], fb[1], 0, lb[1], row, stride); };
            auto ptoken_set = [&]() -> double {
                if (ta == 0 || tb == 0) return 0.0;
                differences();
                if (nc > 0) return 100.0;
                return g_partial_ratio_diff(da, db, row, stride);
            };
            auto ptoken = [&]() -> double {
                if (ta == 0 || tb == 0) return 0.0;
                differences();
                if (nc > 0) return 100.0;
                return fz_fmax(g_partial_ratio(fa[1], fb[1], row, stride), g_partial_ratio_diff(da, db, row, stride));
            };
            double score;
            if (mode == kWRatio) {
                if (la[0] == 0 || lb[0] == 0) score = 0.0;
                else {
                    double end_ratio = g_ratio(fa[0], 0, la[0], fb[0], 0, lb[0], row, stride);
                    const int lmax = fz_max(la[0], lb[0]), lmin = fz_min(la[0], lb[0]);
                    if (2 * lmax < 3 * lmin) score = fz_fmax(end_ratio, fz_fmax(token_sort(), token_set()) * 0.95);
                    else {
                        const double scale = lmax < 8 * lmin ? 0.9 : 0.6;
                        end_ratio = fz_fmax(end_ratio, g_partial_ratio(fa[0], fb[0], row, stride) * scale);
                        score = fz_fmax(end_ratio, ptoken() * 0.95 * scale);
                    }
                }
            }
            else if (mode == kPartialRatio) score = g_partial_ratio(fa[0], fb[0], row, stride);
            else if (mode == kTokenSetRatio) score = token_set();
            else if (mode == kTokenRatio) score = fz_fmax(token_sort(), token_set());
            else if (mode == kPartialTokenSortRatio) score = g_partial_ratio(fa[1], fb[1], row, stride);
            else if (mode == kPartialTokenSetRatio) score = ptoken_set();
            else score = ptoken();
            if (score > best_score || (score == best_score && orig < best_idx)) {
                best_score = score;
                best_idx = orig;
            }
        }
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
            const int64_t o = (int64_t)A.row_slot[r] * A.n_parts_total + A.part0;
            A.part_score[o] = best_score;
            A.part_idx[o] = best_idx;
        }
        __syncthreads();
    }
}

const int32_t *fuzz_forms_tok_pos(const pfz_strings *S);      // k7_fuzz.hip

int fuzz_general_launch(pfz_ctx *ctx, FuzzArgs A, const pfz_strings *F, const pfz_strings *T, const std::vector<int32_t> &rows)
{
    if (rows.empty()) return PFZ_OK;
    int64_t max_a = 1;
    for (int32_t i : rows) max_a = std::max<int64_t>(max_a, F->h_off[(size_t)i + 1] - F->h_off[(size_t)i]);
    const int64_t max_b = std::max<int64_t>(T->max_len, 1);
    if (max_a >= INT_MAX / 2 || max_b >= INT_MAX / 2) {
        set_error("pfz_fuzz: a string of more than 2^30 characters");
        return PFZ_ERR_UNSUPPORTED;
    }
    GeneralScratch S;
    S.max_row = (int32_t)std::max(max_a, max_b);
    S.max_a = (int32_t)max_a;
    S.max_b = (int32_t)max_b;
    // as many workgroups as fit 1 GiB of scratch (at least one)
    const size_t per_wg = ((size_t)S.max_row + 2 + (size_t)S.max_a + 1 + (size_t)S.max_b + 1) * 256 * sizeof(int32_t);
    int64_t grid = std::min<int64_t>((int64_t)rows.size(), (int64_t)ctx->prop.multiProcessorCount * 2);
    while (grid > 1 && per_wg * (size_t)grid > ((size_t)1 << 30)) grid /= 2;
    if (per_wg > ((size_t)8 << 30)) {
        set_error("pfz_fuzz: strings of %lld / %lld characters need %zu bytes of scratch per workgroup", (long long)max_a, (long long)max_b, per_wg);
        return PFZ_ERR_UNSUPPORTED;
    }
    void *d_rows = nullptr, *d_scratch = nullptr;
    PFZ_TRY(pool_alloc_raw(ctx, &d_rows, rows.size() * sizeof(int32_t)));
    struct Free {
        void *p;
        ~Free() { if (p) pool_free(p); }
    } f1{d_rows};
    PFZ_TRY(copy_h2d(ctx, d_rows, rows.data(), rows.size() * sizeof(int32_t)));
    PFZ_TRY(pool_alloc_raw(ctx, &d_scratch, per_wg * (size_t)grid));
    Free f2{d_scratch};
    S.row = (int32_t *)d_scratch;
    S.dpos_a = S.row + (size_t)grid * ((size_t)S.max_row + 2) * 256;
    S.dpos_b = S.dpos_a + (size_t)grid * ((size_t)S.max_a + 1) * 256;
    A.rows = (const int32_t *)d_rows;
    A.n_rows = (int32_t)rows.size();
    hipLaunchKernelGGL(k7_general_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, A, S, fuzz_forms_tok_pos(F));
    PFZ_HIP(hipGetLastError());
    return PFZ_OK;
}

}  // namespace pfz
