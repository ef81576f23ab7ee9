"""Shared helpers of the parity tests (test infrastructure; may use oracle/)."""
import numpy as np

SCORE_TOL = 1e-5      # north-star tolerance for fp32 cosine scores
NEAR_TIE = 2e-6       # float64 oracle scores closer than this may swap ranks in fp32

_cache = {}


def vectorize_pair(oracle, from_list, to_list, cache_key=None, **kw):
    """TF-IDF CSR triples (float64) of from/to with the oracle's vectoriser
    (fit on to+from, reference _tfidf.py:109)."""
    if cache_key is not None and cache_key in _cache:
        return _cache[cache_key]
    v = oracle.TfidfOracle(**kw)
    if to_list is None:
        v.fit(from_list)
        a3 = v.transform(from_list)
        out = (a3, a3, len(v.vocabulary))
    else:
        v.fit(list(to_list) + list(from_list))
        out = (v.transform(from_list), v.transform(to_list), len(v.vocabulary))
    if cache_key is not None:
        _cache[cache_key] = out
    return out


def random_csr(rng, n_rows, n_cols, density, empty_rows=()):
    """Random non-negative CSR with L2-normalised rows, sorted indices."""
    indptr = [0]
    indices, data = [], []
    for r in range(n_rows):
        if r in empty_rows:
            indptr.append(len(indices))
            continue
        k = max(1, int(rng.binomial(n_cols, density)))
        cols = np.sort(rng.choice(n_cols, size=min(k, n_cols), replace=False))
        vals = rng.random(len(cols)) + 0.05
        vals /= np.sqrt((vals * vals).sum())
        indices.extend(cols.tolist())
        data.extend(vals.tolist())
        indptr.append(len(indices))
    return (np.array(indptr, np.int64), np.array(indices, np.int32), np.array(data, np.float64))


def assert_topn_parity(idx, val, exp_idx, exp_val, oracle, a3, b3, n_col, exclude_diag=False,
                       max_near_tie_frac=0.01):
    """idx/val: engine output (int32, fp32); exp_*: oracle (canonical order, float64)."""
    idx = np.asarray(idx)
    val = np.asarray(val, np.float64)
    exp_idx = np.asarray(exp_idx)
    exp_val = np.asarray(exp_val, np.float64)
    assert idx.shape == exp_idx.shape, (idx.shape, exp_idx.shape)
    if idx.size == 0:
        return
    np.testing.assert_allclose(val, exp_val, rtol=0, atol=SCORE_TOL)
    bad_rows = np.nonzero((idx != exp_idx).any(axis=1))[0]
    assert len(bad_rows) <= max(1, int(max_near_tie_frac * len(idx))), \
        f"{len(bad_rows)} of {len(idx)} rows differ from the oracle's indices"
    for i in bad_rows:
        dense = oracle.cossim_dense(a3, b3, n_col, rows=(int(i), int(i) + 1))[0]
        got = idx[i]
        real = got[got >= 0]
        assert len(set(real.tolist())) == len(real), f"row {i}: duplicate indices {got}"
        if exclude_diag:
            assert i not in real.tolist()
        for r in range(idx.shape[1]):
            if got[r] == exp_idx[i, r]:
                continue
            # a swap is only acceptable between candidates the float64 oracle itself
            # separates by less than NEAR_TIE
            s_got = dense[got[r]] if got[r] >= 0 else 0.0
            assert abs(s_got - exp_val[i, r]) < NEAR_TIE, \
                f"row {i} rank {r}: got col {got[r]} (oracle score {s_got!r}), expected col " \
                f"{exp_idx[i, r]} (score {exp_val[i, r]!r})"
