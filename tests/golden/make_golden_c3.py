"""
Config 3 AT FULL SIZE through the REFERENCE's own class (round 6, VERDICT r5 "next" 1b):

    polyfuzz.models.EditDistance(scorer=<restated fuzz.ratio>, normalize=False / True).match(from_list, to_list)

on SURVEY section 8d's 20 000 x 20 000 IMDB titles -- `/root/reference/polyfuzz/models/_distance.py:69-102`: the joblib loop
(`n_jobs=1`, the default), `_calculate_edit_distance`'s `to_list.copy()`, the Python list of 20 000 scores per from-string,
`np.argmax` (FIRST maximum) / `np.max`, the frame, the min-max normalisation.  rapidfuzz is not installable here, so the scorer the
class calls 4 x 10^8 times is the oracle's restatement of `rapidfuzz.fuzz.ratio` (oracle/indel.c) -- served from a per-from-string
row that oracle/indel.c computes in one call (a Python-to-C round trip per PAIR would take hours; the values are the same
function's).  What this pins at full size is therefore the reference's PLUMBING, on the real lists with their 1 300 duplicate
titles and non-ASCII strings; the scorer itself stays "parity unpinned" (DESIGN section 2).

Output: tests/golden/c3_editdistance_golden.npz -- per from-title the To title as the index of its FIRST occurrence in to_list
(the reference returns the string; first occurrence = what np.argmax over the scores picks), the Similarity (float64, the
reference's `value`), and the normalised column.  ~4 min on one core.  python tests/golden/make_golden_c3.py
"""
import json
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import oracle  # noqa: E402


class RowServedRatio:
    """scorer(from_string, to_string) -> oracle/indel.c's fuzz.ratio of the pair; the 20 000 scores of one from-string are
    computed by ONE call into the C restatement when the class asks for the first of them"""

    def __init__(self, to_list):
        self.to_list = list(to_list)
        self.cur = None
        self.row = None
        self.calls = 0

    def __call__(self, a, b, **kw):
        if a != self.cur:
            _, _, mat = oracle.indel_argmax([a], self.to_list, want_matrix=True)
            self.row = dict(zip(self.to_list, mat[0].tolist()))      # (equal strings -> equal scores)
            self.cur = a
        self.calls += 1
        return self.row[b]


def main():
    sys.modules["seaborn"] = types.ModuleType("seaborn")
    rf = types.ModuleType("rapidfuzz")
    rf.fuzz = types.ModuleType("rapidfuzz.fuzz")
    rf.process = types.ModuleType("rapidfuzz.process")
    rf.fuzz.ratio = rf.fuzz.WRatio = lambda a, b, **kw: oracle.indel_ratio(a, b)
    rf.process.extractOne = None
    sys.modules.update({"rapidfuzz": rf, "rapidfuzz.fuzz": rf.fuzz, "rapidfuzz.process": rf.process})
    from polyfuzz.models import EditDistance

    oracle.build_native()
    imdb = json.load(open(os.path.join(REF, "data", "movie_titles.json")))["IMDB"]
    perm = np.random.default_rng(0).permutation(len(imdb))
    fl = [imdb[i] for i in perm[:20000]]
    tl = [imdb[i] for i in perm[20000:40000]]
    assert fl[0] == "Polly Blue Eyes"
    first = {}
    for i, s in enumerate(tl):
        first.setdefault(s, i)
    out = {}
    for norm in (False, True):
        sc = RowServedRatio(tl)
        t0 = time.time()
        df = EditDistance(scorer=sc, normalize=norm).match(fl, tl)
        print(f"reference EditDistance(normalize={norm}).match(): {time.time() - t0:.0f} s, {sc.calls} scorer calls")
        assert sc.calls == len(fl) * len(tl) and df["From"].tolist() == fl
        idx = np.array([first[t] for t in df["To"].tolist()], np.int32)
        if not norm:
            out["idx"], out["score"] = idx, df["Similarity"].to_numpy(np.float64)
        else:
            assert np.array_equal(idx, out["idx"])
            out["normalized"] = df["Similarity"].to_numpy(np.float64)
    np.savez_compressed(os.path.join(HERE, "c3_editdistance_golden.npz"), **out)
    print("wrote c3_editdistance_golden.npz", os.path.getsize(os.path.join(HERE, "c3_editdistance_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
