/*
 * oracle/cossim_topn.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, float64) of the sparse cosine top-n step of the
 * reference's TF-IDF matcher.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may call this; the product path never does.
 *
 * What it restates
 * ----------------
 *  reference polyfuzz/models/_utils.py:73-91  (the "sparse" branch of
 *  cosine_similarity): C = A * B^T through the third-party, un-vendored
 *  sparse_dot_topn.awesome_cossim_topn(A, B.T, top_n+1, min_similarity)
 *  (setup.py:28, extra "fast", sparse_dot_topn>=0.2.9; not installed here,
 *  so its published algorithm is restated): row-by-row Gustavson product with
 *  a dense accumulator over the columns of C, keep per row the entries with
 *  value STRICTLY greater than lower_bound, return the ntop largest.
 *  Then _utils.py:84-87 (self-match: the diagonal is dropped) and
 *  _utils.py:128-146 (_top_n_idx_sparse / _top_n_similarities_sparse: the
 *  top_n column ids and their scores, padded with None/0).
 *
 * Canonical order (the reference leaves ties undefined: np.argpartition /
 * an unstable partial sort): score descending, then column index ascending.
 *
 * Arithmetic: float64, products and sums rounded separately (compile with
 * -ffp-contract=off), per (i,j) the terms are added in ascending n-gram id
 * order starting from +0.0 -- the order scipy's csr_matmat / sparse_dot_topn
 * produce for CSR inputs with sorted indices.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double s; int32_t j; } cand_t;

static int cand_cmp(const void *pa, const void *pb)
{
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->s > b->s) return -1;
    if (a->s < b->s) return 1;
    return (a->j > b->j) - (a->j < b->j);
}

/* Transpose a CSR matrix (n_rows x n_cols) into CSR of its transpose.
 * Rows of the result list the original row ids in ascending order. */
static void csr_transpose(int64_t n_rows, int64_t n_cols,
                          const int64_t *indptr, const int32_t *idx, const double *val,
                          int64_t *t_indptr, int32_t *t_idx, double *t_val)
{
    int64_t nnz = indptr[n_rows];
    memset(t_indptr, 0, (size_t)(n_cols + 1) * sizeof(int64_t));
    for (int64_t p = 0; p < nnz; ++p) t_indptr[idx[p] + 1]++;
    for (int64_t k = 0; k < n_cols; ++k) t_indptr[k + 1] += t_indptr[k];
    int64_t *cur = (int64_t *)malloc((size_t)(n_cols + 1) * sizeof(int64_t));
    memcpy(cur, t_indptr, (size_t)(n_cols + 1) * sizeof(int64_t));
    for (int64_t r = 0; r < n_rows; ++r)
        for (int64_t p = indptr[r]; p < indptr[r + 1]; ++p) {
            int64_t q = cur[idx[p]]++;
            t_idx[q] = (int32_t)r;
            t_val[q] = val[p];
        }
    free(cur);
}

/*
 * A: n_a x n_col CSR (from-side), B: n_b x n_col CSR (to-side), both float64
 * with sorted column indices.  Rows [row_begin, row_end) of A are processed
 * (lets the bench time a bounded sample); out_idx/out_val are
 * (row_end-row_begin) x ntop, row-major; missing entries idx=-1, val=0.
 * exclude_diag != 0: column j == global row i is never a candidate
 * (reference _utils.py:84-87).  Returns 0, or -1 on allocation failure.
 */
static int cossim_topn_core(int64_t n_b, int64_t n_col,
                            const int64_t *a_indptr, const int32_t *a_idx, const double *a_val,
                            const int64_t *b_indptr, const int32_t *b_idx, const double *b_val,
                            int64_t row_begin, int64_t row_end, const int64_t *row_ids,
                            int32_t ntop, double lower_bound, int32_t exclude_diag,
                            int32_t *out_idx, double *out_val)
{
    int64_t nnz_b = b_indptr[n_b];
    int64_t *t_indptr = (int64_t *)malloc((size_t)(n_col + 1) * sizeof(int64_t));
    int32_t *t_idx = (int32_t *)malloc((size_t)(nnz_b > 0 ? nnz_b : 1) * sizeof(int32_t));
    double *t_val = (double *)malloc((size_t)(nnz_b > 0 ? nnz_b : 1) * sizeof(double));
    double *sums = (double *)calloc((size_t)(n_b > 0 ? n_b : 1), sizeof(double));
    int32_t *touched = (int32_t *)malloc((size_t)(n_b > 0 ? n_b : 1) * sizeof(int32_t));
    unsigned char *mark = (unsigned char *)calloc((size_t)(n_b > 0 ? n_b : 1), 1);
    cand_t *cand = (cand_t *)malloc((size_t)(n_b > 0 ? n_b : 1) * sizeof(cand_t));
    if (!t_indptr || !t_idx || !t_val || !sums || !touched || !mark || !cand) return -1;
    csr_transpose(n_b, n_col, b_indptr, b_idx, b_val, t_indptr, t_idx, t_val);

    for (int64_t at = row_begin; at < row_end; ++at) {
        const int64_t i = row_ids ? row_ids[at] : at;      /* the row of A (its global index: the diagonal test below) */
        int64_t n_touched = 0;
        for (int64_t p = a_indptr[i]; p < a_indptr[i + 1]; ++p) {
            int32_t k = a_idx[p];
            double a = a_val[p];
            for (int64_t q = t_indptr[k]; q < t_indptr[k + 1]; ++q) {
                int32_t j = t_idx[q];
                double prod = a * t_val[q];
                sums[j] = sums[j] + prod;
                if (!mark[j]) { mark[j] = 1; touched[n_touched++] = j; }
            }
        }
        int64_t n_cand = 0;
        for (int64_t t = 0; t < n_touched; ++t) {
            int32_t j = touched[t];
            double s = sums[j];
            sums[j] = 0.0;
            mark[j] = 0;
            if (exclude_diag && (int64_t)j == i) continue;
            if (s > lower_bound) { cand[n_cand].s = s; cand[n_cand].j = j; ++n_cand; }
        }
        qsort(cand, (size_t)n_cand, sizeof(cand_t), cand_cmp);
        int32_t *oi = out_idx + (at - row_begin) * ntop;
        double *ov = out_val + (at - row_begin) * ntop;
        for (int32_t r = 0; r < ntop; ++r) {
            if (r < n_cand) { oi[r] = cand[r].j; ov[r] = cand[r].s; }
            else { oi[r] = -1; ov[r] = 0.0; }
        }
    }
    free(t_indptr); free(t_idx); free(t_val); free(sums); free(touched); free(mark); free(cand);
    return 0;
}

int oracle_cossim_topn(int64_t n_a, int64_t n_b, int64_t n_col,
                       const int64_t *a_indptr, const int32_t *a_idx, const double *a_val,
                       const int64_t *b_indptr, const int32_t *b_idx, const double *b_val,
                       int64_t row_begin, int64_t row_end,
                       int32_t ntop, double lower_bound, int32_t exclude_diag,
                       int32_t *out_idx, double *out_val)
{
    (void)n_a;
    return cossim_topn_core(n_b, n_col, a_indptr, a_idx, a_val, b_indptr, b_idx, b_val, row_begin, row_end, NULL, ntop,
                            lower_bound, exclude_diag, out_idx, out_val);
}

/* The same for an arbitrary selection of rows of A (a seeded random sample: bench.py's parity check);
 * out_idx / out_val are n_sel x ntop in the order of row_ids. */
int oracle_cossim_topn_rows(int64_t n_a, int64_t n_b, int64_t n_col,
                            const int64_t *a_indptr, const int32_t *a_idx, const double *a_val,
                            const int64_t *b_indptr, const int32_t *b_idx, const double *b_val,
                            const int64_t *row_ids, int64_t n_sel,
                            int32_t ntop, double lower_bound, int32_t exclude_diag,
                            int32_t *out_idx, double *out_val)
{
    for (int64_t t = 0; t < n_sel; ++t)
        if (row_ids[t] < 0 || row_ids[t] >= n_a) return -2;
    return cossim_topn_core(n_b, n_col, a_indptr, a_idx, a_val, b_indptr, b_idx, b_val, 0, n_sel, row_ids, ntop,
                            lower_bound, exclude_diag, out_idx, out_val);
}

/* Full dense score row(s) for small cases: out is (row_end-row_begin) x n_b. */
int oracle_cossim_dense(int64_t n_b, int64_t n_col,
                        const int64_t *a_indptr, const int32_t *a_idx, const double *a_val,
                        const int64_t *b_indptr, const int32_t *b_idx, const double *b_val,
                        int64_t row_begin, int64_t row_end, double *out)
{
    int64_t nnz_b = b_indptr[n_b];
    int64_t *t_indptr = (int64_t *)malloc((size_t)(n_col + 1) * sizeof(int64_t));
    int32_t *t_idx = (int32_t *)malloc((size_t)(nnz_b > 0 ? nnz_b : 1) * sizeof(int32_t));
    double *t_val = (double *)malloc((size_t)(nnz_b > 0 ? nnz_b : 1) * sizeof(double));
    if (!t_indptr || !t_idx || !t_val) return -1;
    csr_transpose(n_b, n_col, b_indptr, b_idx, b_val, t_indptr, t_idx, t_val);
    for (int64_t i = row_begin; i < row_end; ++i) {
        double *sums = out + (i - row_begin) * n_b;
        for (int64_t j = 0; j < n_b; ++j) sums[j] = 0.0;
        for (int64_t p = a_indptr[i]; p < a_indptr[i + 1]; ++p) {
            int32_t k = a_idx[p];
            double a = a_val[p];
            for (int64_t q = t_indptr[k]; q < t_indptr[k + 1]; ++q) {
                double prod = a * t_val[q];
                sums[t_idx[q]] = sums[t_idx[q]] + prod;
            }
        }
    }
    free(t_indptr); free(t_idx); free(t_val);
    return 0;
}
