// K7, preparation of the lists: the token forms of a list (whitespace tokens sorted / sorted and deduplicated, joined by one
// space: what rapidfuzz's token scorers compare) and the plan of a to-list (alphabet, character classes, token hash table,
// the strings sorted by length in groups of 64: packed records + the summaries sweep 1 bounds every pair with).  Built on
// the device on first use, cached on the pfz_strings handles -- a repeated list costs nothing.  Split from k7_fuzz.hip
// (the match kernel) in round 4; data layout: DESIGN.md section 3, "K7".
// Replaces what rapidfuzz.process.extractOne does per call on the CPU (reference polyfuzz/models/_rapidfuzz.py:99-113).
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
#define FZ_ANY(x) (__ballot(x) != 0ull)
#include "k7_core.h"
#include "k7_plan.h"

void pfz_fuzz_forms_free(pfz_fuzz_forms *f) { delete f; }
void pfz_fuzz_plan_free(pfz_fuzz_plan *p) { delete p; }

namespace pfz {


// ---- forms of a list ---------------------------------------------------------------------------------------------------

// lexicographic order of two tokens of the same string by code unit (= by code point, Python's str order)
__device__ inline int cmp_tokens(const void *chars, int cw, int64_t base, int s1, int l1, int s2, int l2)
{
    const int n = l1 < l2 ? l1 : l2;
    for (int k = 0; k < n; ++k) {
        const uint32_t x = load_unit(chars, cw, base + s1 + k), y = load_unit(chars, cw, base + s2 + k);
        if (x != y) return x < y ? -1 : 1;
    }
    return l1 < l2 ? -1 : (l1 > l2 ? 1 : 0);
}

// one thread per string: split on whitespace, sort the tokens, write " ".join(sorted(tokens)) and
// " ".join(sorted(set(tokens))) with the distinct tokens' positions, lengths and hashes
__global__ __launch_bounds__(128) void k7_tokenize(const void *__restrict__ chars, int cw, const int64_t *__restrict__ off, int64_t n,
                                                    void *__restrict__ form1, void *__restrict__ form2, int32_t *__restrict__ len1,
                                                    int32_t *__restrict__ len2, int32_t *__restrict__ ntok, int32_t *__restrict__ ntok_all,
                                                    int32_t *__restrict__ tok_pos, int32_t *__restrict__ tok_len,
                                                    uint64_t *__restrict__ tok_hash)
{
    const int64_t i = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    const int64_t o = off[i];
    const int len = (int)(off[i + 1] - o);
    const int64_t tb = tok_base(o, i);
    int32_t *ts = tok_pos + tb, *tl = tok_len + tb;       // first: token starts / lengths within the string
    int nt = 0;
    for (int p = 0; p < len;) {
        while (p < len && is_space_cp(load_unit(chars, cw, o + p))) ++p;
        const int b = p;
        while (p < len && !is_space_cp(load_unit(chars, cw, o + p))) ++p;
        if (p > b) {
            ts[nt] = b;
            tl[nt] = p - b;
            ++nt;
        }
    }
    for (int k = 1; k < nt; ++k) {                          // insertion sort (a handful of tokens per string)
        const int ks = ts[k], kl = tl[k];
        int j = k - 1;
        while (j >= 0 && cmp_tokens(chars, cw, o, ts[j], tl[j], ks, kl) > 0) {
            ts[j + 1] = ts[j];
            tl[j + 1] = tl[j];
            --j;
        }
        ts[j + 1] = ks;
        tl[j + 1] = kl;
    }
    int o1 = 0, o2 = 0, nd = 0, prev_s = 0, prev_l = -1;
    for (int k = 0; k < nt; ++k) {
        const int s = ts[k], l = tl[k];                     // (read before entry nd <= k is overwritten below)
        if (k) store_unit(form1, cw, o + o1++, 0x20u);
        for (int q = 0; q < l; ++q) store_unit(form1, cw, o + o1 + q, load_unit(chars, cw, o + s + q));
        o1 += l;
        if (prev_l >= 0 && cmp_tokens(chars, cw, o, prev_s, prev_l, s, l) == 0) continue;      // a repeated token
        prev_s = s;
        prev_l = l;
        if (nd) store_unit(form2, cw, o + o2++, 0x20u);
        uint64_t h = 0xcbf29ce484222325ull;
        for (int q = 0; q < l; ++q) {
            const uint32_t c = load_unit(chars, cw, o + s + q);
            store_unit(form2, cw, o + o2 + q, c);
            h = (h ^ c) * 0x100000001b3ull;
        }
        ts[nd] = o2;
        tl[nd] = l;
        tok_hash[tb + nd] = h;
        o2 += l;
        ++nd;
    }
    len1[i] = o1;
    len2[i] = o2;
    ntok[i] = nd;
    ntok_all[i] = nt;
}

int ensure_forms(pfz_ctx *ctx, pfz_strings *S)
{
    if (S->fuzz_forms) return PFZ_OK;
    Owner<pfz_fuzz_forms, pfz_fuzz_forms_free> f(new pfz_fuzz_forms());
    f->ctx = ctx;
    const size_t units = (size_t)std::max<int64_t>(S->n_units, 1) * (size_t)S->char_width + 16;
    const size_t nn = (size_t)std::max<int64_t>(S->n, 1);
    f->tok_cap = S->n_units / 2 + S->n + 2;
    PFZ_TRY(pool_alloc(ctx, &f->form1, units));
    PFZ_TRY(pool_alloc(ctx, &f->form2, units));
    PFZ_TRY(pool_alloc(ctx, &f->len1, nn * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &f->len2, nn * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &f->ntok, nn * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &f->ntok_all, nn * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &f->tok_pos, (size_t)f->tok_cap * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &f->tok_len, (size_t)f->tok_cap * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &f->tok_hash, (size_t)f->tok_cap * sizeof(uint64_t)));
    f->h_ntok.assign((size_t)S->n, 0);
    if (S->n > 0) {
        {
            ProfScope ps(ctx, "k7_prepare");
            hipLaunchKernelGGL(k7_tokenize, dim3((unsigned)((S->n + 127) / 128)), dim3(128), 0, ctx->stream, S->chars, S->char_width,
                               S->offsets, S->n, f->form1, f->form2, f->len1, f->len2, f->ntok, f->ntok_all, f->tok_pos, f->tok_len,
                               f->tok_hash);
            PFZ_HIP(hipGetLastError());
        }
        PFZ_TRY(copy_d2h(ctx, f->h_ntok.data(), f->ntok, (size_t)S->n * sizeof(int32_t)));
    }
    S->fuzz_forms = f.release();
    return PFZ_OK;
}

// ---- the to-side plan ---------------------------------------------------------------------------------------------------

template <int CW>
__global__ __launch_bounds__(256) void k7_mark_alphabet(const void *__restrict__ chars, int64_t n_units, uint32_t *__restrict__ present)
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

// occurrences of every alphabet symbol in the list (the character classes go to the most frequent symbols first)
__global__ __launch_bounds__(256) void k7_sym_count(const void *__restrict__ chars, int cw, int64_t n_units, const uint16_t *__restrict__ lut,
                                                     uint32_t lut_len, int32_t n_sym1, unsigned int *__restrict__ count)
{
    __shared__ unsigned int h[4096];
    const bool in_lds = n_sym1 <= 4096;
    if (in_lds) {
        for (int t = threadIdx.x; t < n_sym1; t += 256) h[t] = 0u;
        __syncthreads();
    }
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_units; p += (int64_t)gridDim.x * 256) {
        const uint32_t c = load_unit(chars, cw, p);
        const int sy = c < lut_len ? (int)lut[c] : 0;
        if (in_lds) atomicAdd(&h[sy], 1u);
        else atomicAdd(&count[sy], 1u);
    }
    if (in_lds) {
        __syncthreads();
        for (int t = threadIdx.x; t < n_sym1; t += 256)
            if (h[t]) atomicAdd(&count[t], h[t]);
    }
}

struct TokenLists {       // the token arrays of one list's forms
    const void *form2;
    int cw;
    const int64_t *off;
    const int32_t *ntok, *tok_pos, *tok_len;
    const uint64_t *tok_hash;
};

__device__ inline bool same_token(const TokenLists &A, int64_t ia, int64_t ra, const TokenLists &B, int64_t ib, int64_t rb)
{
    if (A.tok_hash[ra] != B.tok_hash[rb] || A.tok_len[ra] != B.tok_len[rb]) return false;
    const int64_t pa = A.off[ia] + A.tok_pos[ra], pb = B.off[ib] + B.tok_pos[rb];
    for (int k = 0; k < A.tok_len[ra]; ++k)
        if (load_unit(A.form2, A.cw, pa + k) != load_unit(B.form2, B.cw, pb + k)) return false;
    return true;
}

// string index of a token reference (binary search over the token bases: tok_base is increasing in i)
__device__ inline int64_t owner_of(const int64_t *off, int64_t n, int64_t ref)
{
    int64_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (tok_base(off[mid], mid) <= ref) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// insert = true: the list's own tokens into its table (id = the representative that got the slot first -- which one is
// a race, that it is ONE per distinct token is not); insert = false: another list's tokens looked up (absent: a
// negative id no other token has)
template <bool INSERT>
__global__ __launch_bounds__(128) void k7_token_ids(TokenLists L, int64_t n, TokenLists T, int64_t n_t, int32_t *__restrict__ table,
                                                     uint32_t mask, int32_t *__restrict__ ids)
{
    const int64_t i = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    const int64_t tb = tok_base(L.off[i], i);
    for (int k = 0; k < L.ntok[i]; ++k) {
        const int64_t ref = tb + k;
        uint32_t h = (uint32_t)(L.tok_hash[ref] ^ (L.tok_hash[ref] >> 32)) & mask;
        int32_t id;
        for (;;) {
            const int32_t cur = INSERT ? atomicCAS(&table[h], -1, (int32_t)ref) : table[h];
            if (cur == -1) {
                id = INSERT ? (int32_t)ref : (int32_t)(-2 - ref);
                break;
            }
            if (INSERT && cur == (int32_t)ref) {
                id = cur;
                break;
            }
            const int64_t owner = owner_of(T.off, n_t, cur);
            if (same_token(L, i, ref, T, owner, cur)) {
                id = cur;
                break;
            }
            h = (h + 1) & mask;
        }
        ids[ref] = id;
    }
}

struct PackArgs {
    const void *form[3];
    int cw;
    const int64_t *off;
    const int32_t *len1, *len2, *ntok, *ntok_all, *tok_len, *tok_id;
    const uint16_t *lut;
    uint32_t lut_len;
    const uint8_t *cls;
    int32_t space_rank, space_class;
    const int32_t *b_orig;
    const int4 *meta3;
    uint16_t *sym;
    uint8_t *tag;
    int32_t *p_tok_id, *p_tok_len;
    int4 *meta, *meta2, *meta4;
    uint4 *hist;
    uint2 *pres;
    int64_t n_slots;
};

// slot (group g, lane l) = to-string b_orig[slot]: everything the match kernel reads of it -- its record (symbols of the
// three forms, tags, tokens) and its summary (lengths, class histogram, token signature, first token ids)
__global__ __launch_bounds__(256) void k7_pack(PackArgs A)
{
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= A.n_slots) return;
    const int64_t g = slot >> 6;
    const int lane = (int)(slot & 63);
    const int32_t j = A.b_orig[slot];
    uint32_t hw[kFuzzHistWords];
#pragma unroll
    for (int d = 0; d < kFuzzHistWords; ++d) hw[d] = 0u;
    if (j < 0) {
        A.meta[slot] = make_int4(0, 0, 0, 0);
        A.meta2[slot] = make_int4(0, 0, 0, -1);
        A.meta4[slot] = make_int4(-1, -1, -1, -1);
        A.hist[(g * 2 + 0) * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
        A.hist[(g * 2 + 1) * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
        A.pres[slot] = make_uint2(0u, 0u);
        return;
    }
    uint32_t pw[2] = {0u, 0u};
    const int4 rec = A.meta3[slot];
    const int64_t o = A.off[j];
    const int len[3] = {(int)(A.off[j + 1] - o), A.len1[j], A.len2[j]};
    const int nt = A.ntok[j], nt_all = A.ntok_all[j];
    const bool with_hist = len[0] + (nt_all > 0 ? nt_all - 1 : 0) <= 255;      // no class counter can pass 255
    int n_space = 0, usum = 0;
    for (int v = 0; v < 3; ++v) {
        uint16_t *dst = A.sym + rec.x + (int64_t)v * rec.w;
        for (int p = 0; p < rec.w; ++p) {
            int sy = 0;
            if (p < len[v]) {
                const uint32_t c = load_unit(A.form[v], A.cw, o + p);
                sy = c < A.lut_len ? (int)A.lut[c] : 0;
                // (no whitespace symbol: the joined forms hold single spaces where the string had any run of them)
                if (v == 0 && sy && !is_space_cp(c)) pw[(sy & 63) >> 5] |= 1u << (sy & 31);
                if (v == 0 && with_hist && sy) {
                    const int cl = A.cls[sy];
                    hw[cl >> 2] += 1u << (8 * (cl & 3));
                    ++usum;
                    n_space += sy == A.space_rank;
                }
            }
            dst[p] = (uint16_t)sy;                 // (the padding of the record is zero: symbol 0 matches nothing)
        }
    }
    if (with_hist) {
        const int extra = (nt_all - 1) - n_space;          // form 1 may hold more joining spaces than the string has
        if (extra > 0) {
            hw[A.space_class >> 2] += (uint32_t)extra << (8 * (A.space_class & 3));
            usum += extra;
        }
    }
    // distinct tokens: ids, lengths, and the tag of every character of form 2
    const int64_t tb = tok_base(o, j);
    uint8_t *tg = A.tag + rec.y;
    int32_t *pid = A.p_tok_id + rec.z, *pln = A.p_tok_len + rec.z;
    uint64_t sig = 0ull;
    int first[4] = {-1, -1, -1, -1};
    int pos = 0;
    for (int t = 0; t < nt; ++t) {
        const int32_t id = A.tok_id[tb + t], l = A.tok_len[tb + t];
        pid[t] = id;
        pln[t] = l;
        if (t < 4) first[t] = id;
        sig |= fz_sig_bit(id);
        for (int q = 0; q < l; ++q) tg[pos + q] = (uint8_t)(t & 31);
        if (t + 1 < nt) tg[pos + l] = (uint8_t)((t & 31) | 0x80);
        pos += l + 1;
    }
    A.meta[slot] = make_int4(len[0], len[1], len[2], nt);
    A.meta2[slot] = make_int4((int)(uint32_t)sig, (int)(uint32_t)(sig >> 32), with_hist ? usum : -1, j);
    A.meta4[slot] = make_int4(first[0], first[1], first[2], first[3]);
    A.hist[(g * 2 + 0) * 64 + lane] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    A.hist[(g * 2 + 1) * 64 + lane] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
    A.pres[slot] = make_uint2(pw[0], pw[1]);
}

template <typename T> static int up(pfz_ctx *ctx, T **dst, const std::vector<T> &v)
{
    PFZ_TRY(pool_alloc(ctx, dst, (v.empty() ? 1 : v.size()) * sizeof(T)));
    if (!v.empty()) PFZ_TRY(copy_h2d(ctx, *dst, v.data(), v.size() * sizeof(T)));
    return PFZ_OK;
}

static TokenLists token_lists(const pfz_strings *S)
{
    const pfz_fuzz_forms *f = S->fuzz_forms;
    return TokenLists{f->form2, S->char_width, S->offsets, f->ntok, f->tok_pos, f->tok_len, f->tok_hash};
}

int build_plan(pfz_ctx *ctx, pfz_strings *T)
{
    PFZ_TRY(ensure_forms(ctx, T));
    const pfz_fuzz_forms *f = T->fuzz_forms;
    Owner<pfz_fuzz_plan, pfz_fuzz_plan_free> pl(new pfz_fuzz_plan());
    pl->ctx = ctx;
    ProfScope ps(ctx, "k7_prepare");
    // alphabet: presence bitmap on the device, ranks on the host; the joining space is always a symbol
    const size_t words = 0x110000 / 32, used_words = T->char_width == 1 ? 8 : words;
    uint32_t *present = nullptr;
    PFZ_TRY(pool_alloc(ctx, &present, words * sizeof(uint32_t)));
    struct Free {
        void *p;
        ~Free() { if (p) pool_free(p); }
    } free_present{present};
    PFZ_HIP(hipMemsetAsync(present, 0, used_words * sizeof(uint32_t), ctx->stream));
    if (T->n_units > 0) {
        const unsigned grid = (unsigned)std::min<int64_t>((T->n_units + 255) / 256, 2048);
        if (T->char_width == 1) hipLaunchKernelGGL(k7_mark_alphabet<1>, dim3(grid), dim3(256), 0, ctx->stream, T->chars, T->n_units, present);
        else hipLaunchKernelGGL(k7_mark_alphabet<4>, dim3(grid), dim3(256), 0, ctx->stream, T->chars, T->n_units, present);
        PFZ_HIP(hipGetLastError());
    }
    std::vector<uint32_t> h(used_words);
    PFZ_TRY(copy_d2h(ctx, h.data(), present, used_words * sizeof(uint32_t)));
    h[0x20 >> 5] |= 1u << (0x20 & 31);
    std::vector<uint16_t> lut;
    std::vector<uint32_t> cp_of_rank(1, 0u);
    for (size_t wi = used_words; wi-- > 0;)
        if (h[wi]) {
            lut.assign((wi + 1) * 32, 0);
            break;
        }
    int32_t S = 0;
    for (size_t wi = 0; wi * 32 < lut.size(); ++wi) {
        uint32_t word = h[wi];
        while (word) {
            const int bit = __builtin_ctz(word);
            word &= word - 1;
            if (S >= 65534) {
                set_error("pfz_fuzz: more than 65534 distinct code points in the to-list exceed the 16-bit symbol space");
                return PFZ_ERR_UNSUPPORTED;
            }
            lut[wi * 32 + (size_t)bit] = (uint16_t)(++S);
            cp_of_rank.push_back((uint32_t)(wi * 32 + (size_t)bit));
        }
    }
    pl->n_sym = S;
    pl->lut_len = (uint32_t)lut.size();
    pl->space_rank = lut[0x20];
    PFZ_TRY(up(ctx, &pl->lut, lut));
    // character classes: symbols in order of decreasing frequency take classes 0, 1, ..., 31, 0, 1, ...
    unsigned int *d_count = nullptr;
    PFZ_TRY(pool_alloc(ctx, &d_count, (size_t)(S + 1) * sizeof(unsigned int)));
    Free free_count{d_count};
    PFZ_HIP(hipMemsetAsync(d_count, 0, (size_t)(S + 1) * sizeof(unsigned int), ctx->stream));
    if (T->n_units > 0) {
        const unsigned grid = (unsigned)std::min<int64_t>((T->n_units + 255) / 256, 1024);
        hipLaunchKernelGGL(k7_sym_count, dim3(grid), dim3(256), 0, ctx->stream, T->chars, T->char_width, T->n_units, pl->lut, pl->lut_len,
                           S + 1, d_count);
        PFZ_HIP(hipGetLastError());
    }
    std::vector<unsigned int> count((size_t)S + 1);
    PFZ_TRY(copy_d2h(ctx, count.data(), d_count, count.size() * sizeof(unsigned int)));
    std::vector<int32_t> order((size_t)S);
    for (int32_t r = 0; r < S; ++r) order[(size_t)r] = r + 1;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return count[(size_t)x] > count[(size_t)y]; });
    std::vector<uint8_t> cls((size_t)S + 1, 0);
    for (int32_t k = 0; k < S; ++k) cls[(size_t)order[(size_t)k]] = (uint8_t)(k % (4 * kFuzzHistWords));
    pl->space_class = cls[(size_t)pl->space_rank];
    PFZ_TRY(up(ctx, &pl->cls, cls));
    // token table of the to-list
    const int64_t n_to = T->n;
    int64_t total_tok = 0;
    for (int64_t j = 0; j < n_to; ++j) total_tok += f->h_ntok[(size_t)j];
    uint32_t cap = 1024;
    while ((int64_t)cap < 2 * total_tok + 16) cap <<= 1;
    pl->table_mask = cap - 1;
    PFZ_TRY(pool_alloc(ctx, &pl->table, (size_t)cap * sizeof(int32_t)));
    PFZ_HIP(hipMemsetAsync(pl->table, 0xff, (size_t)cap * sizeof(int32_t), ctx->stream));
    PFZ_TRY(pool_alloc(ctx, &pl->t_tok_id, (size_t)f->tok_cap * sizeof(int32_t)));
    if (n_to > 0) {
        const TokenLists L = token_lists(T);
        hipLaunchKernelGGL(k7_token_ids<true>, dim3((unsigned)((n_to + 127) / 128)), dim3(128), 0, ctx->stream, L, n_to, L, n_to, pl->table,
                           pl->table_mask, pl->t_tok_id);
        PFZ_HIP(hipGetLastError());
    }
    // groups of 64 to-strings of similar length: counting sort by length on the host (O(n) ints)
    std::vector<int64_t> start((size_t)T->max_len + 2, 0);
    for (int64_t j = 0; j < n_to; ++j) start[(size_t)(T->h_off[(size_t)j + 1] - T->h_off[(size_t)j]) + 1]++;
    for (size_t l = 1; l < start.size(); ++l) start[l] += start[l - 1];
    const int64_t n_groups = (n_to + 63) / 64;
    if (n_groups * 64 > ((int64_t)1 << 26)) {      // (the kernel's window-sweep items hold a to-slot in 26 bits)
        set_error("pfz_fuzz: a to-list of %lld strings exceeds the 2^26-string plan", (long long)n_to);
        return PFZ_ERR_UNSUPPORTED;
    }
    std::vector<int32_t> b_orig((size_t)n_groups * 64, -1);
    for (int64_t j = 0; j < n_to; ++j) {      // ascending j inside one length: a stable sort
        const int64_t len = T->h_off[(size_t)j + 1] - T->h_off[(size_t)j];
        b_orig[(size_t)start[(size_t)len]++] = (int32_t)j;
    }
    // record offsets: no form is longer than the string, a string of len characters has at most (len + 1) / 2 tokens
    const size_t n_slots = (size_t)std::max<int64_t>(n_groups * 64, 1);
    std::vector<int4> meta3(n_slots, make_int4(0, 0, 0, 8));
    int64_t total = 0, tag_total = 0, ttotal = 0;
    for (int64_t sl = 0; sl < n_groups * 64; ++sl) {
        const int32_t j = b_orig[(size_t)sl];
        const int64_t len = j >= 0 ? T->h_off[(size_t)j + 1] - T->h_off[(size_t)j] : 0;
        const int64_t cap8 = std::max<int64_t>(8, (len + 7) & ~(int64_t)7);
        if (total + 3 * cap8 >= INT_MAX || tag_total + cap8 + 16 >= INT_MAX) {
            set_error("pfz_fuzz: a to-list of %lld code units exceeds the 2^31-symbol plan", (long long)T->n_units);
            return PFZ_ERR_UNSUPPORTED;
        }
        meta3[(size_t)sl] = make_int4((int)total, (int)tag_total, (int)ttotal, (int)cap8);
        total += 3 * cap8;
        tag_total += (cap8 + 15) & ~(int64_t)15;
        ttotal += std::max<int64_t>(4, ((len + 1) / 2 + 3) & ~(int64_t)3);
        if (j >= 0 && f->h_ntok[(size_t)j] > kFuzzMaxTokens) pl->big_slots.push_back((int32_t)sl);
    }
    pl->n_groups = n_groups;
    PFZ_TRY(up(ctx, &pl->b_orig, b_orig));
    PFZ_TRY(up(ctx, &pl->meta3, meta3));
    PFZ_TRY(up(ctx, &pl->d_big_slots, pl->big_slots));
    PFZ_TRY(pool_alloc(ctx, &pl->sym, (size_t)(total + 64) * sizeof(uint16_t)));
    PFZ_TRY(pool_alloc(ctx, &pl->tag, (size_t)(tag_total + 64)));
    PFZ_TRY(pool_alloc(ctx, &pl->tok_id, (size_t)(ttotal + 64) * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &pl->tok_len, (size_t)(ttotal + 64) * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &pl->meta, n_slots * sizeof(int4)));
    PFZ_TRY(pool_alloc(ctx, &pl->meta2, n_slots * sizeof(int4)));
    PFZ_TRY(pool_alloc(ctx, &pl->meta4, n_slots * sizeof(int4)));
    PFZ_TRY(pool_alloc(ctx, &pl->hist, n_slots * 2 * sizeof(uint4)));
    PFZ_TRY(pool_alloc(ctx, &pl->pres, n_slots * sizeof(uint2)));
    if (n_groups > 0) {
        PackArgs P;
        P.form[0] = T->chars;
        P.form[1] = f->form1;
        P.form[2] = f->form2;
        P.cw = T->char_width;
        P.off = T->offsets;
        P.len1 = f->len1;
        P.len2 = f->len2;
        P.ntok = f->ntok;
        P.ntok_all = f->ntok_all;
        P.tok_len = f->tok_len;
        P.tok_id = pl->t_tok_id;
        P.lut = pl->lut;
        P.lut_len = pl->lut_len;
        P.cls = pl->cls;
        P.space_rank = pl->space_rank;
        P.space_class = pl->space_class;
        P.b_orig = pl->b_orig;
        P.meta3 = pl->meta3;
        P.sym = pl->sym;
        P.tag = pl->tag;
        P.p_tok_id = pl->tok_id;
        P.p_tok_len = pl->tok_len;
        P.meta = pl->meta;
        P.meta2 = pl->meta2;
        P.meta4 = pl->meta4;
        P.hist = pl->hist;
        P.pres = pl->pres;
        P.n_slots = n_groups * 64;
        hipLaunchKernelGGL(k7_pack, dim3((unsigned)((n_groups * 64 + 255) / 256)), dim3(256), 0, ctx->stream, P);
        PFZ_HIP(hipGetLastError());
    }
    T->fuzz_plan = pl.release();
    return PFZ_OK;
}

int from_token_ids(pfz_ctx *ctx, const pfz_strings *F, const pfz_strings *T, int32_t *d_out)
{
    const pfz_fuzz_plan *pl = T->fuzz_plan;
    hipLaunchKernelGGL(k7_token_ids<false>, dim3((unsigned)((F->n + 127) / 128)), dim3(128), 0, ctx->stream, token_lists(F), F->n,
                       token_lists(T), T->n, pl->table, pl->table_mask, d_out);
    PFZ_HIP(hipGetLastError());
    return PFZ_OK;
}

}  // namespace pfz
