"""
oracle/reference_path.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference's own EXECUTABLE TF-IDF path, restated step by step with the very
third-party calls it makes -- the CPU arm "(i)" of SURVEY.md §8d and the thing
`tests/golden/make_golden.py` ran through the real package.  /root/reference does
not exist on the GPU box, scikit-learn does, so bench.py's cpu_baseline leg times
THIS (kind "port" of a path whose arithmetic is sklearn's own):

* reference polyfuzz/models/_tfidf.py:102-118  TfidfVectorizer(min_df=1,
  analyzer=_create_ngrams).fit(to + from) / .transform(...)   (sklearn itself)
* reference polyfuzz/models/_utils.py:54-56    top_n clipped to len(set(to_list))
* reference polyfuzz/models/_utils.py:94-102   the branch that runs whenever
  sparse_dot_topn is missing (or cosine_method="sklearn"): dense
  sklearn.metrics.pairwise.cosine_similarity, fill_diagonal(0) for a self-match,
  full argsort + sort per row, flipped, first top_n, np.round(..., 3)
* reference polyfuzz/models/_utils.py:104-125  frame: object vstack, astype(float),
  `< 0.001 -> 0 / None`

Pinned by tests/test_oracle_cpu.py against the frames the reference itself
produced (tests/golden/readme_cases.json).
"""
import time

import numpy as np
import pandas as pd

from .tfidf_oracle import create_ngrams


def sklearn_backend_match(from_list, to_list=None, top_n=1, n_gram_range=(3, 3), clean_string=True,
                          remove_space_ngrams=True, timings=None):
    """-> DataFrame exactly as `TFIDF(cosine_method="sklearn", ...).match(from_list, to_list)` builds it."""
    from sklearn.feature_extraction.text import TfidfVectorizer
    from sklearn.metrics.pairwise import cosine_similarity as sk_cosine

    def analyzer(s):
        return create_ngrams(s, n_gram_range, clean_string, remove_space_ngrams)

    t0 = time.perf_counter()
    vec = TfidfVectorizer(min_df=1, analyzer=analyzer)
    if to_list is not None:
        vec.fit(list(to_list) + list(from_list))
        a, b = vec.transform(from_list), vec.transform(to_list)
        top_n = min(top_n, len(set(to_list)))
    else:
        vec.fit(from_list)
        a = b = vec.transform(from_list)
    t1 = time.perf_counter()
    sim = sk_cosine(a, b)
    if to_list is None:
        np.fill_diagonal(sim, 0)
    order = np.flip(np.argsort(sim, axis=-1), axis=1)[:, :top_n]
    best = np.flip(np.sort(sim, axis=-1), axis=1)[:, :top_n]
    rounded = [np.round(best[:, r], 3) for r in range(best.shape[1])]
    t2 = time.perf_counter()
    names = list(from_list) if to_list is None else to_list
    cols = (["From"] + ["To" if r == 0 else f"To_{r + 1}" for r in range(top_n)] +
            ["Similarity" if r == 0 else f"Similarity_{r + 1}" for r in range(top_n)])
    picked = [[names[j] for j in order[:, r]] for r in range(order.shape[1])]
    df = pd.DataFrame(np.vstack(([from_list], picked, rounded)).T, columns=cols)
    df = df.loc[:, ["From", "To", "Similarity"] + [c for r in range(1, top_n) for c in (f"To_{r + 1}", f"Similarity_{r + 1}")]]
    for c in df.columns:
        if "Similarity" in c:
            df[c] = df[c].astype(float)
            low = df[c] < 0.001
            df.loc[low, c] = float(0)
            df.loc[low, c.replace("Similarity", "To")] = None
    t3 = time.perf_counter()
    if timings is not None:
        timings.update({"vectorise_s": t1 - t0, "cosine_sort_s": t2 - t1, "frame_s": t3 - t2, "total_s": t3 - t0})
    return df


def rapidfuzz_shared_list_self_match(names, scorer="WRatio", score_cutoff=0.0, use_c=True):
    """The reference RapidFuzz matcher's self-match as its code runs it with n_jobs = 1 (polyfuzz/models/_rapidfuzz.py:86-113):
    ONE copy of the list, `to_list.remove(from_string)` before every `process.extractOne(from_string, to_list, score_cutoff=
    self.score_cutoff, scorer=self.scorer)` -- restated literally, list.remove included, so that what is left of the list is
    whatever Python leaves.  Returns (From, To (None = no match), Similarity = score / 100), lists of len(names).
    score_cutoff is the constructor's 0..1 value (the class multiplies by 100).  use_c: the scorers of fuzz_scorers.c
    (== fuzz_scorers.py bit for bit) instead of the Python ones -- the same numbers, fast enough for a few thousand names."""
    from . import fuzz_scorers, native
    to_list = list(names)
    cut = score_cutoff * 100
    frm, to, sim = [], [], []
    for s in names:
        to_list.remove(s)
        best_j, best = -1, 0.0
        if to_list:
            if use_c:
                idx, score = native.fuzz_extract_one([s], to_list, scorer)
                best_j, best = int(idx[0]), float(score[0])
            else:
                idx, score = fuzz_scorers.extract_one_all([s], to_list, fuzz_scorers.SCORERS[scorer])
                best_j, best = idx[0], score[0]
        frm.append(s)
        if best_j >= 0 and best >= cut:            # extractOne: the first best choice, None below score_cutoff
            to.append(to_list[best_j])
            sim.append(best / 100)
        else:
            to.append(None)
            sim.append(0.0)
    return frm, to, sim
