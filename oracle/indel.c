/*
 * oracle/indel.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's EditDistance matcher hot loop:
 *   polyfuzz/models/_distance.py:89-102 (_calculate_edit_distance): score one
 *   from-string against every to-string with scorer = rapidfuzz.fuzz.ratio
 *   (_distance.py:4,32), np.argmax (FIRST maximum) and np.max.
 *
 * rapidfuzz (setup.py:20, rapidfuzz>=0.13.1, C++, un-vendored and NOT
 * installed in this image -> PARITY UNPINNED for the scorer itself against an
 * executable reference; anchored on the values rapidfuzz publishes, see
 * tests/test_oracle_cpu.py::test_indel_ratio_published_known_answers) publishes
 * fuzz.ratio as the normalised Indel similarity * 100:
 *     dist      = |a| + |b| - 2 * LCS(a, b)          (insert/delete only)
 *     norm_dist = (|a|+|b| == 0) ? 0.0 : dist / (|a|+|b|)      (float64)
 *     ratio     = (1.0 - norm_dist) * 100.0
 * on sequences of Unicode code points.  The LCS here is the plain O(|a||b|)
 * dynamic programme -- deliberately NOT the bit-parallel algorithm the HIP
 * kernel uses, so the two are independent.
 *
 * Self-match (_distance.py:70-73, 93-96): the to-list is a copy of the
 * from-list with the FIRST element equal to the from-string removed
 * (list.remove), which is not always element i when the list has duplicates.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int64_t lcs_len(const uint32_t *a, int64_t la, const uint32_t *b, int64_t lb, int32_t *row)
{
    /* row has lb+1 entries */
    for (int64_t j = 0; j <= lb; ++j) row[j] = 0;
    for (int64_t i = 0; i < la; ++i) {
        int32_t diag = 0;               /* L[i][0] */
        for (int64_t j = 0; j < lb; ++j) {
            int32_t up = row[j + 1];    /* L[i][j+1]   */
            int32_t v;
            if (a[i] == b[j]) v = diag + 1;
            else v = (up > row[j]) ? up : row[j];
            diag = up;
            row[j + 1] = v;
        }
    }
    return row[lb];
}

double oracle_indel_ratio(const uint32_t *a, int64_t la, const uint32_t *b, int64_t lb)
{
    int32_t *row = (int32_t *)malloc((size_t)(lb + 1) * sizeof(int32_t));
    int64_t lcs = lcs_len(a, la, b, lb, row);
    free(row);
    int64_t maximum = la + lb;
    int64_t dist = maximum - 2 * lcs;
    double norm_dist = (maximum != 0) ? (double)dist / (double)maximum : 0.0;
    return (1.0 - norm_dist) * 100.0;
}

/*
 * All-pairs best match.  Strings are uint32 code points, string i of list X
 * is X_cp[X_off[i] .. X_off[i+1]).  Rows [row_begin,row_end) of the from-list
 * are scored.  self_match != 0 restates list.remove(from_string): the first
 * to-entry equal to the from-string is skipped.  out_idx = index into the
 * ORIGINAL to-list of the first maximum (-1 if the candidate list is empty),
 * out_score = the float64 ratio.  opt_matrix (may be NULL) receives the full
 * (row_end-row_begin) x n_b score matrix (skipped entry = -1).
 */
int oracle_indel_argmax(const uint32_t *a_cp, const int64_t *a_off, int64_t n_a,
                        const uint32_t *b_cp, const int64_t *b_off, int64_t n_b,
                        int64_t row_begin, int64_t row_end, int32_t self_match,
                        int32_t *out_idx, double *out_score, double *opt_matrix)
{
    (void)n_a;
    int64_t max_lb = 0;
    for (int64_t j = 0; j < n_b; ++j)
        if (b_off[j + 1] - b_off[j] > max_lb) max_lb = b_off[j + 1] - b_off[j];
    int32_t *row = (int32_t *)malloc((size_t)(max_lb + 1) * sizeof(int32_t));
    if (!row) return -1;
    for (int64_t i = row_begin; i < row_end; ++i) {
        const uint32_t *a = a_cp + a_off[i];
        int64_t la = a_off[i + 1] - a_off[i];
        int64_t skip = -1;
        if (self_match) {
            for (int64_t j = 0; j < n_b; ++j) {
                int64_t lb = b_off[j + 1] - b_off[j];
                if (lb == la && memcmp(b_cp + b_off[j], a, (size_t)la * sizeof(uint32_t)) == 0) { skip = j; break; }
            }
        }
        int32_t best = -1;
        double best_s = 0.0;
        for (int64_t j = 0; j < n_b; ++j) {
            if (j == skip) { if (opt_matrix) opt_matrix[(i - row_begin) * n_b + j] = -1.0; continue; }
            const uint32_t *b = b_cp + b_off[j];
            int64_t lb = b_off[j + 1] - b_off[j];
            int64_t lcs = lcs_len(a, la, b, lb, row);
            int64_t maximum = la + lb;
            int64_t dist = maximum - 2 * lcs;
            double norm_dist = (maximum != 0) ? (double)dist / (double)maximum : 0.0;
            double s = (1.0 - norm_dist) * 100.0;
            if (opt_matrix) opt_matrix[(i - row_begin) * n_b + j] = s;
            if (best < 0 || s > best_s) { best = (int32_t)j; best_s = s; }
        }
        out_idx[i - row_begin] = best;
        out_score[i - row_begin] = best_s;
    }
    free(row);
    return 0;
}
