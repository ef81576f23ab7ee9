"""End-to-end parity of the drop-in matchers (polyfuzz_amd.models.TFIDF / EditDistance)
against DataFrames produced by the REFERENCE itself (tests/golden/, make_golden.py)."""
import io
import pickle

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def _cmp_frame(df, rec, sim_atol=1.01e-3):
    assert list(df.columns) == list(rec.keys())
    for c in df.columns:
        got = df[c].tolist()
        if "Similarity" in c:
            np.testing.assert_allclose(np.array(got, float), np.array(rec[c], float), atol=sim_atol)
        else:
            assert [None if (isinstance(v, float) and v != v) else v for v in got] == rec[c], c


def test_readme_cases_match_reference_frames(golden):
    from polyfuzz_amd.models import TFIDF
    rc = golden["readme_cases"]
    fl, tl = rc["from_list"], rc["to_list"]
    n = 0
    for case in rc["cases"]:
        kw = dict(case["kwargs"])
        if "n_gram_range" in kw:
            kw["n_gram_range"] = tuple(kw["n_gram_range"])
        m = TFIDF(cosine_method="sklearn", **kw)      # goldens were made with the reference's sklearn back-end
        df = m.match(fl) if case["self"] else m.match(fl, tl)
        _cmp_frame(df, case["df"])
        n += 1
    assert n >= 18


def test_readme_sparse_semantics_min_similarity():
    from polyfuzz_amd.models import TFIDF
    fl = ["apple", "apples", "appl", "recal", "house", "similarity"]
    tl = ["apple", "apples", "mouse"]
    df = TFIDF().match(fl, tl)      # defaults: min_similarity 0.75, "sparse" -> strict lower bound honoured
    assert df["To"].tolist() == ["apple", "apples", "apple", None, None, None]
    np.testing.assert_allclose(df["Similarity"], [1.0, 1.0, 0.784, 0, 0, 0], atol=1e-9)
    df = TFIDF(min_similarity=0, top_n=2).match(fl)      # self-match, SURVEY §8c
    assert df["To"].tolist()[:3] == ["apples", "apple", "apple"]
    np.testing.assert_allclose(df["Similarity"][:3], [0.787, 0.787, 0.767], atol=1e-9)
    np.testing.assert_allclose(df["Similarity_2"][:3], [0.767, 0.604, 0.604], atol=1e-9)


def test_fit_transform_and_pickle(golden):
    from polyfuzz_amd.models import TFIDF
    rc = golden["readme_cases"]
    fl, tl = rc["from_list"], rc["to_list"]
    m = TFIDF(cosine_method="sklearn", min_similarity=0, top_n=1)
    m.match(fl, tl)
    new_from = rc["transform"]["new_from"]
    _cmp_frame(m.match(new_from, tl, re_train=False), rc["transform"]["df"])
    assert m.vectorizer.get_feature_names_out()[0] == "app" and m.tf_idf_to.shape == (3, 19)
    m2 = pickle.loads(pickle.dumps(m))
    _cmp_frame(m2.match(new_from, tl, re_train=False), rc["transform"]["df"])
    assert m2.vectorizer.vocabulary_ == m.vectorizer.vocabulary_


def test_company_c2_frame_vs_reference(golden):
    from polyfuzz_amd.models import TFIDF
    fl = golden["company_c2_lists"]["from_list"]
    tl = golden["company_c2_lists"]["to_list"]
    df = TFIDF(cosine_method="sklearn", min_similarity=0, top_n=5).match(fl, tl)
    ref_sim = golden["npz"]["c2_ref_sim"]
    canon = golden["npz"]["c2_canon_idx"]
    pos = {}
    for i, s in enumerate(tl):
        pos.setdefault(s, i)
    bad = 0
    for r in range(5):
        sc = "Similarity" if r == 0 else f"Similarity_{r + 1}"
        tc = "To" if r == 0 else f"To_{r + 1}"
        np.testing.assert_allclose(df[sc].to_numpy(), ref_sim[:, r], atol=1.01e-3)
        got = np.array([-1 if t is None else pos[t] for t in df[tc].tolist()])
        exp = np.where(ref_sim[:, r] < 0.001, -1, canon[:, r])
        bad += int((got != exp).sum())
    assert bad <= 0.002 * 5 * len(fl)      # only near-ties in the reference's own float64 scores may differ


def test_edit_distance_frames(golden):
    from polyfuzz_amd.models import EditDistance
    rc = golden["readme_cases"]
    fl, tl = rc["from_list"], rc["to_list"]
    for case in rc["edit_distance"]:
        m = EditDistance(normalize=case["normalize"])
        df = m.match(fl) if case["self"] else m.match(fl, tl)
        _cmp_frame(df, case["df"], sim_atol=1e-12)
    t = golden["titles_lists"]
    for norm in (True, False):
        df = EditDistance(normalize=norm).match(t["from_list"], t["to_list"])
        np.testing.assert_allclose(df["Similarity"].to_numpy(), golden["npz"][f"titles_sim_norm{int(norm)}"], atol=1e-12)
        assert df["To"].tolist() == [t["to_list"][j] for j in golden["npz"][f"titles_idx_norm{int(norm)}"]]
    with pytest.raises(NotImplementedError):
        EditDistance(scorer=lambda a, b: 0.0)


def test_cosine_similarity_operator_sparse_input(oracle_mod):
    from scipy.sparse import csr_matrix
    from polyfuzz_amd.models import cosine_similarity
    from tests.helpers import vectorize_pair
    fl = ["apple", "apples", "appl", "recal", "house", "similarity"]
    tl = ["apple", "apples", "mouse"]
    a3, b3, n_col = vectorize_pair(oracle_mod, fl, tl)
    A = csr_matrix((a3[2], a3[1], a3[0]), shape=(6, n_col))
    B = csr_matrix((b3[2], b3[1], b3[0]), shape=(3, n_col))
    df = cosine_similarity(A, B, fl, tl, min_similarity=0, top_n=5)
    assert list(df.columns) == ["From", "To", "Similarity", "To_2", "Similarity_2", "To_3", "Similarity_3"]
    assert df["To"].tolist() == ["apple", "apples", "apple", None, "mouse", None]


def _frame_idx(df, to_list, top_n):
    pos = {}
    for i, s in enumerate(to_list):
        pos.setdefault(s, i)
    idx = np.full((len(df), top_n), -1, np.int64)
    sim = np.zeros((len(df), top_n))
    for r in range(top_n):
        tc = "To" if r == 0 else f"To_{r + 1}"
        sc = "Similarity" if r == 0 else f"Similarity_{r + 1}"
        idx[:, r] = [(-1 if t is None else pos[t]) for t in df[tc].tolist()]
        sim[:, r] = df[sc].to_numpy()
    return idx, sim


def test_company_self_match_frame_vs_reference(golden):
    """TFIDF.match(list) -- the docs' own use case (self-match, diagonal excluded) -- vs the reference run."""
    from polyfuzz_amd.models import TFIDF
    sl = golden["company_self_list"]["from_list"]
    df = TFIDF(cosine_method="sklearn", min_similarity=0, top_n=3).match(sl)
    idx, sim = _frame_idx(df, sl, 3)
    np.testing.assert_allclose(sim, golden["npz"]["self_ref_sim"], atol=1.01e-3)
    canon = np.where(golden["npz"]["self_ref_sim"] < 0.001, -1, golden["npz"]["self_canon_idx"])
    # duplicate strings in the list make To -> index ambiguous: compare through the strings
    names = np.array(sl + [None], dtype=object)
    same = (names[idx] == names[canon])
    assert (~same).sum() <= 0.003 * same.size


@pytest.mark.parametrize("clean", [True, False])
def test_non_ascii_titles_vs_reference(golden, clean):
    """Movie titles with Latin-1 and CJK / Hangul characters: clean_string=True goes through the host-side
    Unicode-aware cleaning + 1-byte device path, clean_string=False through the UTF-32 device path with an
    alphabet of > 255 symbols.  Goldens: the reference's TFIDF(cosine_method='sklearn')."""
    from polyfuzz_amd.models import TFIDF
    t = golden["titles_lists"]
    fl, tl = t["from_list"], t["to_list"]
    df = TFIDF(cosine_method="sklearn", min_similarity=0, top_n=2, clean_string=clean).match(fl, tl)
    key = "titles_tfidf_clean" if clean else "titles_tfidf_raw"
    idx, sim = _frame_idx(df, tl, 2)
    np.testing.assert_allclose(sim, golden["npz"][key + "_sim"], atol=1.01e-3)
    ref_idx = golden["npz"][key + "_idx"]
    names = np.array(tl + [None], dtype=object)
    diff = names[idx] != names[ref_idx]
    # the reference's order among equal scores is undefined (flipped argsort): only tied ranks may differ
    for i, r in zip(*np.nonzero(diff)):
        row_sims = golden["npz"][key + "_sim"][i]
        assert (np.abs(row_sims - row_sims[r]) < 1.5e-3).sum() >= 2 or row_sims[r] < 0.001, (i, r, df.iloc[i].tolist())


def test_knn_backend_name_vs_reference_frames(golden):
    """cosine_method="knn" (reference _utils.py:59-70) is an alias of the HIP op that ignores min_similarity:
    same frames as the reference's sklearn back-end produced (the reference's knn and sklearn branches agree
    on these lists)."""
    from polyfuzz_amd.models import TFIDF
    rc = golden["readme_cases"]
    fl, tl = rc["from_list"], rc["to_list"]
    for case in rc["cases"][:6]:
        kw = dict(case["kwargs"])
        kw["min_similarity"] = 0.99            # must be ignored by "knn"
        m = TFIDF(cosine_method="knn", **kw)
        df = m.match(fl) if case["self"] else m.match(fl, tl)
        _cmp_frame(df, case["df"])


def test_self_match_job_equals_matcher_and_oracle(ctx, oracle_mod, golden):
    """pipeline.TfidfMatchJob(self_match=True): the fit runs on the list ALONE (reference _tfidf.py:113-116;
    n_docs = len(list), not twice that), single-list and row-shard forms give the oracle's result."""
    from polyfuzz_amd import pipeline
    sl = golden["company_self_list"]["from_list"]
    n = len(sl)
    a3, _, n_col = __import__("tests.helpers", fromlist=["x"]).vectorize_pair(oracle_mod, sl, None)
    e_idx, e_val = oracle_mod.cossim_topn(a3, a3, n_col, 3, 0.0, exclude_diag=True)
    job = pipeline.TfidfMatchJob(ctx, sl, None, top_n=3, min_similarity=0.0, self_match=True)
    idx, val = job.step().download()
    assert job.vec.info()["n_docs"] == n
    assert np.abs(val - e_val).max() <= 1e-5 and (idx != e_idx).any(axis=1).sum() <= 2
    # a row shard of the same job: rows [b, e) against the whole list, diagonal at the shard offset
    b, e = pipeline.shard_bounds(n, 3, 1)
    shard = pipeline.TfidfMatchJob(ctx, sl[b:e], sl, top_n=3, min_similarity=0.0, self_match=True, shard_offset=b)
    s_idx, s_val = shard.step().download()
    np.testing.assert_array_equal(s_idx, idx[b:e])
    np.testing.assert_array_equal(s_val, val[b:e])
