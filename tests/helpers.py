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


# ---- the headline held to the REFERENCE's own run (tests/golden/headline_knn_golden.npz, make_golden_headline.py) -----------

def load_headline_golden():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "headline_knn_golden.npz"))
    return {k: g[k] for k in ("idx", "sim", "sim3", "to_none")}


def assert_topn_equals_reference_knn(idx, val, golden, true_score, rows=None, score_tol=SCORE_TOL):
    """idx / val: an engine's (or the oracle's) self-match top-5 of the 100 000 company names, rows `rows` (default: all), canonical
    order; golden: what `polyfuzz.models.TFIDF(min_similarity=0, top_n=5, cosine_method="knn").match(names)` -- the REFERENCE, run in the
    build container -- returned for them.  true_score(r, j) -> float64 cosine of the pairs (r[i], j[i]) from the oracle-built matrix.

    The rule (SURVEY section 7 "Ties"): the scores agree rank by rank within the north-star's 1e-5 on EVERY cell; wherever the index
    at a rank differs, the reference's choice must score the same as ours at that rank (NEAR_TIE: an exact tie, resolved by
    sklearn's unspecified argpartition order there and by ascending index here) -- or be the row itself, `_utils.py:61-65`'s quirk:
    neighbour column 0 is dropped as "self", which keeps the row's own index wherever an exact duplicate took column 0; such a
    row must have a perfect match here.  Empty cells (score 0 / index -1 here) must be the frame's None cells."""
    idx, val = np.asarray(idx), np.asarray(val, np.float64)
    if rows is None:
        rows = np.arange(len(idx))
    rows = np.asarray(rows)
    r_idx, r_sim, r_none = golden["idx"][rows], golden["sim"][rows].astype(np.float64), golden["to_none"][rows]
    assert idx.shape == r_idx.shape
    err = np.abs(val - r_sim)
    assert err.max() <= score_tol, f"max |score - reference| = {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"
    empty = idx < 0
    assert np.array_equal(empty, val == 0)
    # the reference's None cells are its cells below 0.001 after rounding: ours that round the same way
    mine_none = np.round(val, 3) < 0.001
    edge = np.abs(r_sim - 0.0005) < 2 * score_tol
    assert np.array_equal(mine_none | edge, r_none | edge)
    rr, kk = np.nonzero((idx != r_idx) & ~r_none & ~empty)
    jj = r_idx[rr, kk]
    is_self = jj == rows[rr]
    ts = true_score(rows[rr[~is_self]], jj[~is_self])
    not_tie = np.abs(ts - val[rr[~is_self], kk[~is_self]]) > NEAR_TIE
    assert not not_tie.any(), (f"{int(not_tie.sum())} cells where the reference chose a column that is no tie of ours; first: row "
                               f"{rows[rr[~is_self]][not_tie][:3]}, reference column {jj[~is_self][not_tie][:3]}")
    assert (val[rr[is_self], 0] >= 1.0 - NEAR_TIE).all() and (val[rr[is_self], kk[is_self]] >= 1.0 - NEAR_TIE).all()
    # an empty cell of ours facing a real cell of the reference (or the other way round) was caught by the None test above
    return {"cells": int(idx.size), "cells_index_differs": int(len(rr)), "of_them_reference_kept_self": int(is_self.sum()),
            "max_abs_score_err": float(err.max())}
