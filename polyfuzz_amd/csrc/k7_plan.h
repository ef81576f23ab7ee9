// K7: what the match kernels (k7_fuzz.hip, k7_general.hip) and the preparation of the lists (k7_plan.hip) share -- the
// token forms of a list, the plan of a to-list, and the few helpers both sides read code units and token slots with.
// (reference polyfuzz/models/_rapidfuzz.py:99-113: the lists process.extractOne walks; see k7_fuzz.hip)
#pragma once
#include "pfz_internal.h"

namespace pfz {

__device__ inline uint32_t load_unit(const void *p, int width, int64_t i)
{
    return width == 1 ? (uint32_t)((const uint8_t *)p)[i] : ((const uint32_t *)p)[i];
}

__device__ inline void store_unit(void *p, int width, int64_t i, uint32_t c)
{
    if (width == 1) ((uint8_t *)p)[i] = (uint8_t)c;
    else ((uint32_t *)p)[i] = c;
}

// str.isspace(): what str.split() splits on (the oracle restates rapidfuzz's tokenisation with Python's)
__device__ inline bool is_space_cp(uint32_t c)
{
    return (c >= 0x09u && c <= 0x0Du) || (c >= 0x1Cu && c <= 0x20u) || c == 0x85u || c == 0xA0u || c == 0x1680u ||
           (c >= 0x2000u && c <= 0x200Au) || c == 0x2028u || c == 0x2029u || c == 0x202Fu || c == 0x205Fu || c == 0x3000u;
}

// distinct tokens of string i live at [tok_base(i), tok_base(i) + ntok[i]) of the token arrays (a string of len
// code units has at most (len + 1) / 2 tokens)
__host__ __device__ inline int64_t tok_base(int64_t off_i, int64_t i) { return (off_i >> 1) + i; }


}  // namespace pfz

struct pfz_fuzz_forms {
    pfz_ctx *ctx = nullptr;
    void *form1 = nullptr, *form2 = nullptr;      // code units of the list's width; string i at offsets[i], lengths below
    int32_t *len1 = nullptr, *len2 = nullptr;     // [n]
    int32_t *ntok = nullptr, *ntok_all = nullptr; // [n] distinct tokens / tokens
    int32_t *tok_pos = nullptr, *tok_len = nullptr;   // [tok_cap] start within form 2, length
    uint64_t *tok_hash = nullptr;                 // [tok_cap] FNV-1a over the code points
    int64_t tok_cap = 0;
    std::vector<int32_t> h_ntok;                  // host copy (which from-strings fit the 32-token kernels)
    ~pfz_fuzz_forms()
    {
        for (void *p : {form1, form2, (void *)len1, (void *)len2, (void *)ntok, (void *)ntok_all, (void *)tok_pos, (void *)tok_len,
                        (void *)tok_hash})
            if (p) pfz::pool_free(p);
    }
};

struct pfz_fuzz_plan {
    pfz_ctx *ctx = nullptr;
    int32_t n_sym = 0, space_rank = 0, space_class = 0;
    uint32_t lut_len = 0;
    uint16_t *lut = nullptr;             // code unit -> rank (0: not in the alphabet)
    uint8_t *cls = nullptr;              // [n_sym + 1] rank -> character class (0 .. 31)
    int32_t *table = nullptr;            // token hash table: slot -> token reference (index into the forms' token arrays), -1 empty
    uint32_t table_mask = 0;
    int32_t *t_tok_id = nullptr;         // [tok_cap of the to-list's forms] id of every distinct token (its table representative)
    int64_t n_groups = 0;
    int32_t *b_orig = nullptr;           // [n_groups * 64] original index, -1 = padding lane
    // every to-string's record, contiguous and 16-byte aligned (a lane fetches 8 symbols / 8 tags / 4 tokens per load):
    uint16_t *sym = nullptr;             // forms 0, 1, 2 at meta3.x + v * meta3.w, each padded to meta3.w = pad8(len0) symbols
    uint8_t *tag = nullptr;              // form 2's tags at meta3.y (padded to 16)
    int32_t *tok_id = nullptr, *tok_len = nullptr;    // distinct tokens at meta3.z (padded to 4)
    int4 *meta = nullptr;                // [n_groups * 64] {len0, len1, len2, distinct tokens}
    int4 *meta2 = nullptr;               // [n_groups * 64] {signature lo, hi, histogram sum (-1: none), original index}
    int4 *meta3 = nullptr;               // [n_groups * 64] {symbol offset, tag offset, token offset, padded form length}
    int4 *meta4 = nullptr;               // [n_groups * 64] ids of the first four distinct tokens (-1: none)
    uint4 *hist = nullptr;               // [n_groups][2][64]
    uint2 *pres = nullptr;               // [n_groups * 64] symbol presence (fz_presence_miss)
    std::vector<int32_t> big_slots;      // to-strings with more than 32 distinct tokens (scored by the general kernel)
    int32_t *d_big_slots = nullptr;
    ~pfz_fuzz_plan()
    {
        for (void *p : {(void *)lut, (void *)cls, (void *)table, (void *)t_tok_id, (void *)b_orig, (void *)sym, (void *)tag, (void *)tok_id,
                        (void *)tok_len, (void *)meta, (void *)meta2, (void *)meta3, (void *)meta4, (void *)hist, (void *)d_big_slots})
            if (p) pfz::pool_free(p);
        if (pres) pfz::pool_free(pres);
    }
};


namespace pfz {

// k7_plan.hip -- all three build on first use and cache on the pfz_strings handle
int ensure_forms(pfz_ctx *ctx, pfz_strings *S);      // the token forms of a list (form1 / form2, distinct tokens, hashes)
int build_plan(pfz_ctx *ctx, pfz_strings *T);        // the to-side plan: alphabet, token table, length-sorted packed records + summaries
// ids of the from-list's distinct tokens in the to-list's token table (-1: a token no to-string has) into d_out[tok_cap of F]
int from_token_ids(pfz_ctx *ctx, const pfz_strings *F, const pfz_strings *T, int32_t *d_out);

}  // namespace pfz
