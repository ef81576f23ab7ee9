"""
Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE
(/root/reference, PolyFuzz v0.4.3) in the build container.  The GPU box has no
/root/reference, so the outputs are committed; this script is how they were
made (python tests/golden/make_golden.py).

The reference imports only after two absent third-party modules are stubbed:
* seaborn   (plots only, polyfuzz/metrics.py:3)            -> empty stub
* rapidfuzz (polyfuzz/models/_distance.py:4, _rapidfuzz.py:3) -> stub whose
  fuzz.ratio is the oracle's restatement.  EditDistance goldens therefore pin
  the reference's PLUMBING (arg-max, list.remove self-match, min-max
  normalisation) but NOT the scorer -> "parity unpinned" for rapidfuzz.

TF-IDF goldens come from the reference's own executable back-end here:
TFIDF(cosine_method="sklearn") (sparse_dot_topn is not installed, so "sparse"
silently runs the same branch, _utils.py:8-12,94).
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import oracle  # noqa: E402


def _install_stubs():
    sns = types.ModuleType("seaborn")
    sys.modules["seaborn"] = sns
    rf = types.ModuleType("rapidfuzz")
    fuzz = types.ModuleType("rapidfuzz.fuzz")
    process = types.ModuleType("rapidfuzz.process")
    fuzz.ratio = lambda a, b, **kw: oracle.indel_ratio(a, b)
    fuzz.WRatio = fuzz.ratio
    process.extractOne = lambda *a, **k: None
    rf.fuzz = fuzz
    rf.process = process
    sys.modules["rapidfuzz"] = rf
    sys.modules["rapidfuzz.fuzz"] = fuzz
    sys.modules["rapidfuzz.process"] = process


_install_stubs()
from polyfuzz.models import TFIDF, EditDistance  # noqa: E402
from polyfuzz.models._utils import cosine_similarity as ref_cosine_similarity  # noqa: E402
from sklearn.metrics.pairwise import cosine_similarity as sk_cos  # noqa: E402


def df_to_records(df):
    out = {}
    for c in df.columns:
        col = df[c].tolist()
        out[c] = [None if (v is None or (isinstance(v, float) and v != v)) else v for v in col]
    return out


def frame_to_idx(df, to_list, top_n):
    """To-columns -> index into to_list (-1 for None); Similarity columns -> float64."""
    pos = {}
    for i, s in enumerate(to_list):
        pos.setdefault(s, i)
    idx = np.full((len(df), top_n), -1, np.int32)
    sim = np.zeros((len(df), top_n), np.float64)
    for r in range(top_n):
        tc = "To" if r == 0 else f"To_{r + 1}"
        sc = "Similarity" if r == 0 else f"Similarity_{r + 1}"
        idx[:, r] = [(-1 if t is None else pos[t]) for t in df[tc].tolist()]
        sim[:, r] = df[sc].to_numpy(np.float64)
    return idx, sim


def canonical_topn(dense, top_n, exclude_diag):
    d = dense.copy()
    if exclude_diag:
        np.fill_diagonal(d, -1.0)
    n, m = d.shape
    idx = np.empty((n, top_n), np.int32)
    val = np.empty((n, top_n), np.float64)
    for i in range(n):
        order = np.lexsort((np.arange(m), -d[i]))[:top_n]
        idx[i] = order
        val[i] = d[i, order]
    idx[val <= 0.0] = -1
    val[val <= 0.0] = 0.0
    return idx, val


def main():
    names = json.load(open(os.path.join(REF, "data", "company_names.json")))
    movies = json.load(open(os.path.join(REF, "data", "movie_titles.json")))
    g = {}

    # ---- README lists (reference tests/utils.py:1-4) -----------------------
    fl = ["apple", "apples", "appl", "recal", "house", "similarity"]
    tl = ["apple", "apples", "mouse"]
    readme = {"from_list": fl, "to_list": tl, "cases": []}
    for kw in [dict(min_similarity=0, top_n=1), dict(min_similarity=0, top_n=3),
               dict(min_similarity=0, top_n=5), dict(min_similarity=0.75, top_n=1)]:
        m = TFIDF(cosine_method="sklearn", **kw)
        readme["cases"].append({"kwargs": kw, "self": False, "df": df_to_records(m.match(fl, tl))})
    for kw in [dict(min_similarity=0, top_n=1), dict(min_similarity=0, top_n=2)]:
        m = TFIDF(cosine_method="sklearn", **kw)
        readme["cases"].append({"kwargs": kw, "self": True, "df": df_to_records(m.match(fl))})
    for rng in [(1, 1), (1, 2), (1, 3), (2, 2), (2, 3), (3, 3), (3, 6)]:
        for clean in (True, False):
            kw = dict(min_similarity=0, top_n=2, n_gram_range=rng, clean_string=clean)
            m = TFIDF(cosine_method="sklearn", **kw)
            readme["cases"].append({"kwargs": kw, "self": False, "df": df_to_records(m.match(fl, tl))})
    # fit / transform (re_train=False): reference polyfuzz.py:234-240
    m = TFIDF(cosine_method="sklearn", min_similarity=0, top_n=1)
    m.match(fl, tl)
    new_from = ["apples", "mouses", "zzz", "appl e"]
    readme["transform"] = {"new_from": new_from,
                           "df": df_to_records(m.match(new_from, tl, re_train=False))}
    # EditDistance plumbing (scorer = restated ratio)
    ed = []
    for norm in (True, False):
        ed.append({"normalize": norm, "self": False,
                   "df": df_to_records(EditDistance(normalize=norm).match(fl, tl))})
        ed.append({"normalize": norm, "self": True,
                   "df": df_to_records(EditDistance(normalize=norm).match(fl))})
    readme["edit_distance"] = ed
    json.dump(readme, open(os.path.join(HERE, "readme_cases.json"), "w"), indent=1)

    # ---- company names, config C2 of SURVEY.md §8d -------------------------
    perm = np.random.default_rng(0).permutation(100000)
    c_from = [names[i] for i in perm[:10000]]
    c_to = [names[i] for i in perm[10000:20000]]
    json.dump({"from_list": c_from, "to_list": c_to},
              open(os.path.join(HERE, "company_c2_lists.json"), "w"))
    m = TFIDF(cosine_method="sklearn", min_similarity=0, top_n=5)
    df = m.match(c_from, c_to)
    ref_idx, ref_sim = frame_to_idx(df, c_to, 5)
    # un-rounded scores of the same run: the reference's vectoriser + sklearn dense cosine
    tf_from, tf_to = m._extract_tf_idf(c_from, c_to, re_train=True)
    dense = sk_cos(tf_from, tf_to)
    can_idx, can_val = canonical_topn(dense, 5, False)
    g["c2_ref_idx"] = ref_idx          # as the reference printed them (its own tie order)
    g["c2_ref_sim"] = ref_sim          # rounded to 3 dp by the reference
    g["c2_canon_idx"] = can_idx        # (score desc, col asc) over the reference's dense matrix
    g["c2_canon_val"] = can_val        # un-rounded float64
    g["c2_vocab_size"] = np.array([tf_to.shape[1]], np.int64)
    g["c2_nnz"] = np.array([tf_from.nnz, tf_to.nnz], np.int64)

    # ---- company names self-match (3000 names, top-3) ----------------------
    s_list = [names[i] for i in perm[20000:23000]]
    json.dump({"from_list": s_list}, open(os.path.join(HERE, "company_self_list.json"), "w"))
    m = TFIDF(cosine_method="sklearn", min_similarity=0, top_n=3)
    df = m.match(s_list)
    ref_idx, ref_sim = frame_to_idx(df, s_list, 3)
    tf_from, tf_to = m._extract_tf_idf(s_list, None, re_train=True)
    dense = sk_cos(tf_from, tf_to)
    can_idx, can_val = canonical_topn(dense, 3, True)
    g["self_ref_idx"], g["self_ref_sim"] = ref_idx, ref_sim
    g["self_canon_idx"], g["self_canon_val"] = can_idx, can_val

    # ---- movie titles (non-ASCII incl.), EditDistance plumbing -------------
    imdb = movies["IMDB"]
    netflix = movies["Netflix"]
    p2 = np.random.default_rng(1).permutation(len(imdb))
    non_ascii = [t for t in netflix if not t.isascii()][:20]
    t_from = [imdb[i] for i in p2[:280]] + non_ascii
    t_to = [imdb[i] for i in p2[280:560]] + non_ascii[:10] + [imdb[p2[0]]]
    json.dump({"from_list": t_from, "to_list": t_to},
              open(os.path.join(HERE, "titles_lists.json"), "w"))
    for norm in (True, False):
        df = EditDistance(normalize=norm).match(t_from, t_to)
        idx, sim = frame_to_idx(df, t_to, 1)
        g[f"titles_idx_norm{int(norm)}"] = idx[:, 0]
        g[f"titles_sim_norm{int(norm)}"] = sim[:, 0]
    dup = t_from[:150] + t_from[:20]            # duplicates: list.remove removes the FIRST equal
    json.dump({"from_list": dup}, open(os.path.join(HERE, "titles_self_list.json"), "w"))
    df = EditDistance(normalize=False).match(dup)
    g["titles_self_to"] = np.array([("" if t is None else t) for t in df["To"].tolist()])
    g["titles_self_sim"] = df["Similarity"].to_numpy(np.float64)
    # TF-IDF on titles without cleaning (non-ASCII symbols, case kept)
    m = TFIDF(cosine_method="sklearn", min_similarity=0, top_n=2, clean_string=False)
    df = m.match(t_from, t_to)
    idx, sim = frame_to_idx(df, t_to, 2)
    g["titles_tfidf_raw_idx"], g["titles_tfidf_raw_sim"] = idx, sim
    m = TFIDF(cosine_method="sklearn", min_similarity=0, top_n=2, clean_string=True)
    df = m.match(t_from, t_to)
    idx, sim = frame_to_idx(df, t_to, 2)
    g["titles_tfidf_clean_idx"], g["titles_tfidf_clean_sim"] = idx, sim

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **g)
    print("wrote", sorted(g))


if __name__ == "__main__":
    main()
