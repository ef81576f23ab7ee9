"""
oracle/dense.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

float64 restatement of the dense branch of the reference's cosine_similarity operator
(polyfuzz/models/_utils.py:94-102 -> sklearn.metrics.pairwise.cosine_similarity,
sklearn/metrics/pairwise.py:1683+: normalize(X) . normalize(Y)^T with zero rows left
at zero), followed by the canonical top-n (score desc, column asc), scores > lower_bound.
normalize=False: the raw dot products the "sparse" branch forms from dense input
(_utils.py:74-82: csr_matrix(ndarray) @ csr_matrix(ndarray).T; sparse_dot_topn is not
installable here, so this variant is anchored on the call site, not on a reference run).
"""
import numpy as np


def dense_cossim(a, b, normalize=True):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if not normalize:
        return a @ b.T
    na = np.sqrt((a * a).sum(1))
    nb = np.sqrt((b * b).sum(1))
    na[na == 0] = 1.0
    nb[nb == 0] = 1.0
    return (a / na[:, None]) @ (b / nb[:, None]).T


def _topn_rows(d, ntop, lb, idx, val, row0, select):
    n, m = d.shape
    for i in range(n):
        if select and m > 4 * ntop:
            # the same canonical order from a pre-selection: every entry >= the ntop-th largest value is a candidate (ties of that
            # value included), the full sort runs over the candidates only
            kth = np.partition(d[i], m - ntop)[m - ntop]
            cand = np.flatnonzero(d[i] >= kth)
            order = cand[np.lexsort((cand, -d[i, cand]))][:ntop]
        else:
            order = np.lexsort((np.arange(m), -d[i]))[:ntop]
        keep = d[i, order] > lb
        k = int(keep.sum())
        idx[row0 + i, :k] = order[:k]
        val[row0 + i, :k] = d[i, order[:k]]


def dense_cossim_topn(a, b, ntop, lower_bound=0.0, exclude_diag=False, normalize=True, chunk_rows=None):
    """chunk_rows: None = one dense matrix and a full sort per row (the plain statement); an int = the same result from row chunks of
    that many from-rows with a partition-based pre-selection per row (what makes a thousand rows against 500 000 vectors affordable;
    held equal to the plain statement by tests/test_oracle_cpu.py)."""
    n = len(a)
    idx = np.full((n, ntop), -1, np.int32)
    val = np.zeros((n, ntop), np.float64)
    lb = max(lower_bound, 0.0)
    if chunk_rows is None:
        d = dense_cossim(a, b, normalize)
        if exclude_diag:
            np.fill_diagonal(d, -np.inf)
        _topn_rows(d, ntop, lb, idx, val, 0, False)
        return idx, val
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if normalize:
        na = np.sqrt((a * a).sum(1))
        nb = np.sqrt((b * b).sum(1))
        na[na == 0] = 1.0
        nb[nb == 0] = 1.0
        a = a / na[:, None]
        b = b / nb[:, None]
    bt = b.T
    for r0 in range(0, n, chunk_rows):
        d = a[r0:r0 + chunk_rows] @ bt
        if exclude_diag:
            for i in range(len(d)):
                if r0 + i < d.shape[1]:
                    d[i, r0 + i] = -np.inf
        _topn_rows(d, ntop, lb, idx, val, r0, True)
    return idx, val
