/*
 * oracle/fuzz_scorers.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The rapidfuzz.fuzz scorers the reference's RapidFuzz matcher can be given (polyfuzz/models/_rapidfuzz.py:45-58,
 * 106-108; default fuzz.WRatio) and process.extractOne's first-best rule, in plain C: the same statement as
 * oracle/fuzz_scorers.py, function for function (tests/test_fuzz_oracle_cpu.py holds the two equal bit for bit), fast
 * enough to check thousands of from-rows of the 20 000 x 20 000 title lists and to serve as bench.py's CPU arm for the
 * RapidFuzz configuration.
 *
 * PARITY UNPINNED, like the Python file: rapidfuzz (setup.py:20, `rapidfuzz>=0.13.1`, un-vendored) is not installable
 * here; this restates the published semantics of rapidfuzz 3.x (no default processor; whitespace tokens as Python's
 * str.split(); " ".join(sorted(...)); the three-part partial_ratio window sweep; WRatio's 1.5 / 8 length-ratio
 * branches and 0.95 / 0.9 / 0.6 scales) and is anchored on the values rapidfuzz publishes.
 *
 * Every LCS is the plain O(|a||b|) dynamic programme -- deliberately NOT the bit-parallel algorithm of the HIP
 * kernels -- and every window of partial_ratio is scored on its own, none skipped.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { S_RATIO = 0, S_QRATIO = 1, S_PARTIAL = 2, S_TOKEN_SORT = 3, S_TOKEN_SET = 4, S_TOKEN = 5, S_PTOKEN_SORT = 6,
       S_PTOKEN_SET = 7, S_PTOKEN = 8, S_WRATIO = 9, S_COUNT = 10 };

typedef struct {
    const uint32_t *p;
    int64_t len;
} str_t;

/* one list element: the string, " ".join(sorted(s.split())), sorted(set(s.split())) */
typedef struct {
    str_t s;
    str_t sorted_join;
    int64_t n_tok;          /* distinct tokens, sorted */
    str_t *tok;
    uint32_t *own;          /* storage of sorted_join */
} prep_t;

typedef struct {
    int32_t *row;           /* DP row, grown on demand */
    int64_t row_cap;
    uint32_t *buf_a, *buf_b;   /* joined token differences */
    int64_t buf_cap;
} work_t;

static int64_t lcs_dp(const uint32_t *a, int64_t la, const uint32_t *b, int64_t lb, work_t *w)
{
    if (la == 0 || lb == 0) return 0;
    if (lb + 1 > w->row_cap) {
        w->row_cap = 2 * (lb + 1);
        w->row = (int32_t *)realloc(w->row, (size_t)w->row_cap * sizeof(int32_t));
    }
    int32_t *row = w->row;
    for (int64_t j = 0; j <= lb; ++j) row[j] = 0;
    for (int64_t i = 0; i < la; ++i) {
        int32_t diag = 0;
        for (int64_t j = 0; j < lb; ++j) {
            const int32_t up = row[j + 1];
            row[j + 1] = a[i] == b[j] ? diag + 1 : (up > row[j] ? up : row[j]);
            diag = up;
        }
    }
    return row[lb];
}

static double ratio_of(int64_t dist, int64_t lensum)
{
    const double norm_dist = lensum ? (double)dist / (double)lensum : 0.0;
    return (1.0 - norm_dist) * 100.0;
}

static double norm_distance(int64_t dist, int64_t lensum)          /* rapidfuzz's norm_distance<100> (token_set_ratio) */
{
    return lensum ? 100.0 - (double)(100 * dist) / (double)lensum : 100.0;
}

static double ratio(str_t a, str_t b, work_t *w)
{
    return ratio_of(a.len + b.len - 2 * lcs_dp(a.p, a.len, b.p, b.len, w), a.len + b.len);
}

static double dmax(double x, double y) { return x > y ? x : y; }

/* len(s1) <= len(s2): prefixes of s2 shorter than s1, its windows of length len(s1), its suffixes shorter than s1 */
static double partial_impl(str_t s1, str_t s2, work_t *w)
{
    double best = 0.0;
    for (int64_t i = 1; i < s1.len; ++i) best = dmax(best, ratio(s1, (str_t){s2.p, i}, w));
    for (int64_t i = 0; i < s2.len - s1.len; ++i) best = dmax(best, ratio(s1, (str_t){s2.p + i, s1.len}, w));
    for (int64_t i = s2.len - s1.len; i < s2.len; ++i) best = dmax(best, ratio(s1, (str_t){s2.p + i, s2.len - i}, w));
    return best;
}

static double partial_ratio(str_t s1, str_t s2, work_t *w)
{
    if (s1.len == 0 || s2.len == 0) return s1.len == 0 && s2.len == 0 ? 100.0 : 0.0;
    const str_t shorter = s1.len <= s2.len ? s1 : s2, longer = s1.len <= s2.len ? s2 : s1;
    double res = partial_impl(shorter, longer, w);
    if (res != 100.0 && s1.len == s2.len) res = dmax(res, partial_impl(longer, shorter, w));
    return res;
}

static int tok_cmp(str_t x, str_t y)          /* Python's str ordering: code points, a proper prefix first */
{
    const int64_t n = x.len < y.len ? x.len : y.len;
    for (int64_t i = 0; i < n; ++i)
        if (x.p[i] != y.p[i]) return x.p[i] < y.p[i] ? -1 : 1;
    return x.len < y.len ? -1 : (x.len > y.len ? 1 : 0);
}

static int tok_cmp_q(const void *x, const void *y) { return tok_cmp(*(const str_t *)x, *(const str_t *)y); }

static int is_space(uint32_t c)               /* str.isspace(): what str.split() splits on */
{
    return (c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x20) || c == 0x85 || c == 0xA0 || c == 0x1680 ||
           (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

static void prepare(prep_t *e, const uint32_t *p, int64_t len)
{
    e->s = (str_t){p, len};
    str_t *tok = (str_t *)malloc((size_t)(len / 2 + 1) * sizeof(str_t));
    int64_t n = 0;
    for (int64_t i = 0; i < len;) {
        while (i < len && is_space(p[i])) ++i;
        const int64_t b = i;
        while (i < len && !is_space(p[i])) ++i;
        if (i > b) tok[n++] = (str_t){p + b, i - b};
    }
    qsort(tok, (size_t)n, sizeof(str_t), tok_cmp_q);
    e->own = (uint32_t *)malloc((size_t)(len + 1) * sizeof(uint32_t));
    int64_t o = 0;
    for (int64_t t = 0; t < n; ++t) {
        if (t) e->own[o++] = ' ';
        memcpy(e->own + o, tok[t].p, (size_t)tok[t].len * sizeof(uint32_t));
        o += tok[t].len;
    }
    e->sorted_join = (str_t){e->own, o};
    int64_t d = 0;
    for (int64_t t = 0; t < n; ++t)
        if (d == 0 || tok_cmp(tok[d - 1], tok[t]) != 0) tok[d++] = tok[t];
    e->n_tok = d;
    e->tok = tok;
}

static void release(prep_t *e)
{
    free(e->tok);
    free(e->own);
}

/* the sorted-joined tokens of x that are not in y; *common += tokens in both (counted once, from x's side) */
static str_t joined_difference(const prep_t *x, const prep_t *y, uint32_t *buf, int64_t *n_common, int64_t *sect_len)
{
    int64_t o = 0, j = 0, nc = 0, sl = 0, first = 1, first_c = 1;
    for (int64_t i = 0; i < x->n_tok; ++i) {
        while (j < y->n_tok && tok_cmp(y->tok[j], x->tok[i]) < 0) ++j;
        if (j < y->n_tok && tok_cmp(y->tok[j], x->tok[i]) == 0) {
            ++nc;
            sl += x->tok[i].len + (first_c ? 0 : 1);
            first_c = 0;
            continue;
        }
        if (!first) buf[o++] = ' ';
        first = 0;
        memcpy(buf + o, x->tok[i].p, (size_t)x->tok[i].len * sizeof(uint32_t));
        o += x->tok[i].len;
    }
    if (n_common) *n_common = nc;
    if (sect_len) *sect_len = sl;
    return (str_t){buf, o};
}

static void need_bufs(work_t *w, int64_t n)
{
    if (n + 1 > w->buf_cap) {
        w->buf_cap = 2 * (n + 1);
        w->buf_a = (uint32_t *)realloc(w->buf_a, (size_t)w->buf_cap * sizeof(uint32_t));
        w->buf_b = (uint32_t *)realloc(w->buf_b, (size_t)w->buf_cap * sizeof(uint32_t));
    }
}

static double token_sort_ratio(const prep_t *a, const prep_t *b, work_t *w) { return ratio(a->sorted_join, b->sorted_join, w); }

static double token_set_ratio(const prep_t *a, const prep_t *b, work_t *w)
{
    if (a->n_tok == 0 || b->n_tok == 0) return 0.0;
    need_bufs(w, a->s.len > b->s.len ? a->s.len : b->s.len);
    int64_t n_common = 0, sect_len = 0;
    const str_t diff_ab = joined_difference(a, b, w->buf_a, &n_common, &sect_len);
    const str_t diff_ba = joined_difference(b, a, w->buf_b, NULL, NULL);
    if (n_common > 0 && (diff_ab.len == 0 || diff_ba.len == 0)) return 100.0;     /* (a difference of >= 1 token is never empty) */
    const int64_t ab_len = diff_ab.len, ba_len = diff_ba.len, sep = sect_len != 0;
    const int64_t sect_ab_len = sect_len + sep + ab_len, sect_ba_len = sect_len + sep + ba_len;
    const int64_t dist = ab_len + ba_len - 2 * lcs_dp(diff_ab.p, ab_len, diff_ba.p, ba_len, w);
    const double result = norm_distance(dist, sect_ab_len + sect_ba_len);
    if (!sect_len) return result;
    const double sect_ab_ratio = norm_distance(sep + ab_len, sect_len + sect_ab_len);
    const double sect_ba_ratio = norm_distance(sep + ba_len, sect_len + sect_ba_len);
    return dmax(result, dmax(sect_ab_ratio, sect_ba_ratio));
}

static double partial_token_sort_ratio(const prep_t *a, const prep_t *b, work_t *w)
{
    return partial_ratio(a->sorted_join, b->sorted_join, w);
}

static double partial_token_set_ratio(const prep_t *a, const prep_t *b, work_t *w)
{
    if (a->n_tok == 0 || b->n_tok == 0) return 0.0;
    need_bufs(w, a->s.len > b->s.len ? a->s.len : b->s.len);
    int64_t n_common = 0;
    const str_t diff_ab = joined_difference(a, b, w->buf_a, &n_common, NULL);
    if (n_common > 0) return 100.0;
    const str_t diff_ba = joined_difference(b, a, w->buf_b, NULL, NULL);
    return partial_ratio(diff_ab, diff_ba, w);
}

static double partial_token_ratio(const prep_t *a, const prep_t *b, work_t *w)
{
    if (a->n_tok == 0 || b->n_tok == 0) return 0.0;
    int64_t n_common = 0;
    need_bufs(w, a->s.len > b->s.len ? a->s.len : b->s.len);
    joined_difference(a, b, w->buf_a, &n_common, NULL);
    if (n_common > 0) return 100.0;
    return dmax(partial_token_sort_ratio(a, b, w), partial_token_set_ratio(a, b, w));
}

static double wratio(const prep_t *a, const prep_t *b, work_t *w)
{
    const double UNBASE_SCALE = 0.95;
    if (a->s.len == 0 || b->s.len == 0) return 0.0;
    const double len1 = (double)a->s.len, len2 = (double)b->s.len;
    const double len_ratio = len1 > len2 ? len1 / len2 : len2 / len1;
    double end_ratio = ratio(a->s, b->s, w);
    if (len_ratio < 1.5)
        return dmax(end_ratio, dmax(token_sort_ratio(a, b, w), token_set_ratio(a, b, w)) * UNBASE_SCALE);
    const double PARTIAL_SCALE = len_ratio < 8.0 ? 0.9 : 0.6;
    end_ratio = dmax(end_ratio, partial_ratio(a->s, b->s, w) * PARTIAL_SCALE);
    return dmax(end_ratio, partial_token_ratio(a, b, w) * UNBASE_SCALE * PARTIAL_SCALE);
}

static double score_pair(const prep_t *a, const prep_t *b, int32_t scorer, work_t *w)
{
    switch (scorer) {
    case S_RATIO: return ratio(a->s, b->s, w);
    case S_QRATIO: return a->s.len == 0 || b->s.len == 0 ? 0.0 : ratio(a->s, b->s, w);
    case S_PARTIAL: return partial_ratio(a->s, b->s, w);
    case S_TOKEN_SORT: return token_sort_ratio(a, b, w);
    case S_TOKEN_SET: return token_set_ratio(a, b, w);
    case S_TOKEN: return dmax(token_sort_ratio(a, b, w), token_set_ratio(a, b, w));
    case S_PTOKEN_SORT: return partial_token_sort_ratio(a, b, w);
    case S_PTOKEN_SET: return partial_token_set_ratio(a, b, w);
    case S_PTOKEN: return partial_token_ratio(a, b, w);
    default: return wratio(a, b, w);
    }
}

static void work_free(work_t *w)
{
    free(w->row);
    free(w->buf_a);
    free(w->buf_b);
}

double oracle_fuzz_score(const uint32_t *a, int64_t la, const uint32_t *b, int64_t lb, int32_t scorer)
{
    if (scorer < 0 || scorer >= S_COUNT) return -1.0;
    prep_t pa, pb;
    work_t w = {0};
    prepare(&pa, a, la);
    prepare(&pb, b, lb);
    const double s = score_pair(&pa, &pb, scorer, &w);
    release(&pa);
    release(&pb);
    work_free(&w);
    return s;
}

/* process.extractOne for from-rows [row_begin, row_end): the FIRST choice with the highest score; skip[i] (or -1) is a
 * choice index left out for from-string i.  out_idx = -1 / out_score = 0 when there is no choice. */
int oracle_fuzz_extract_one(const uint32_t *a_cp, const int64_t *a_off, int64_t n_a, const uint32_t *b_cp, const int64_t *b_off,
                            int64_t n_b, int32_t scorer, const int32_t *skip, int64_t row_begin, int64_t row_end,
                            int32_t *out_idx, double *out_score)
{
    if (scorer < 0 || scorer >= S_COUNT || row_begin < 0 || row_end > n_a || row_begin > row_end) return 1;
    prep_t *pb = (prep_t *)malloc((size_t)(n_b > 0 ? n_b : 1) * sizeof(prep_t));
    if (!pb) return 2;
    for (int64_t j = 0; j < n_b; ++j) prepare(&pb[j], b_cp + b_off[j], b_off[j + 1] - b_off[j]);
    work_t w = {0};
    for (int64_t i = row_begin; i < row_end; ++i) {
        prep_t pa;
        prepare(&pa, a_cp + a_off[i], a_off[i + 1] - a_off[i]);
        int64_t best_j = -1;
        double best = -1.0;
        for (int64_t j = 0; j < n_b; ++j) {
            if (skip && j == skip[i]) continue;
            const double v = score_pair(&pa, &pb[j], scorer, &w);
            if (v > best) {
                best_j = j;
                best = v;
            }
        }
        out_idx[i - row_begin] = (int32_t)best_j;
        out_score[i - row_begin] = best_j >= 0 ? best : 0.0;
        release(&pa);
    }
    for (int64_t j = 0; j < n_b; ++j) release(&pb[j]);
    free(pb);
    work_free(&w);
    return 0;
}

/* every score of from-rows [row_begin, row_end) x all choices, row-major (tests of the kernels' per-pair arithmetic) */
int oracle_fuzz_matrix(const uint32_t *a_cp, const int64_t *a_off, int64_t n_a, const uint32_t *b_cp, const int64_t *b_off,
                       int64_t n_b, int32_t scorer, int64_t row_begin, int64_t row_end, double *out)
{
    if (scorer < 0 || scorer >= S_COUNT || row_begin < 0 || row_end > n_a || row_begin > row_end) return 1;
    prep_t *pb = (prep_t *)malloc((size_t)(n_b > 0 ? n_b : 1) * sizeof(prep_t));
    if (!pb) return 2;
    for (int64_t j = 0; j < n_b; ++j) prepare(&pb[j], b_cp + b_off[j], b_off[j + 1] - b_off[j]);
    work_t w = {0};
    for (int64_t i = row_begin; i < row_end; ++i) {
        prep_t pa;
        prepare(&pa, a_cp + a_off[i], a_off[i + 1] - a_off[i]);
        for (int64_t j = 0; j < n_b; ++j) out[(i - row_begin) * n_b + j] = score_pair(&pa, &pb[j], scorer, &w);
        release(&pa);
    }
    for (int64_t j = 0; j < n_b; ++j) release(&pb[j]);
    free(pb);
    work_free(&w);
    return 0;
}
