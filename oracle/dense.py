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


def dense_cossim_topn(a, b, ntop, lower_bound=0.0, exclude_diag=False, normalize=True):
    d = dense_cossim(a, b, normalize)
    if exclude_diag:
        np.fill_diagonal(d, -np.inf)
    n, m = d.shape
    idx = np.full((n, ntop), -1, np.int32)
    val = np.zeros((n, ntop), np.float64)
    lb = max(lower_bound, 0.0)
    for i in range(n):
        order = np.lexsort((np.arange(m), -d[i]))[:ntop]
        keep = d[i, order] > lb
        k = int(keep.sum())
        idx[i, :k] = order[:k]
        val[i, :k] = d[i, order[:k]]
    return idx, val
