"""
Golden fixtures of the two reductions on the hot path's output -- single_linkage and
precision_recall_curve -- made by RUNNING THE REFERENCE functions (/root/reference,
polyfuzz/linkage.py, polyfuzz/metrics.py; pure Python, importable here once seaborn /
rapidfuzz are stubbed).  Output: tests/golden/group_golden.json.  The dicts are stored as
item lists: their insertion ORDER is part of the reference's behaviour.

Frames:
* "readme"  : the reference's own test frame, TFIDF(cosine_method=...).match(from_list, to_list)
              (tests/test_linkage.py:8-10), thresholds 0 .. 1 as in its parametrisation;
* "self"    : self-match top-1 of 2 000 company names -- what PolyFuzz._create_groups feeds
              single_linkage (polyfuzz.py:474-475) -- rebuilt from the reference-run goldens of
              tests/golden/golden.npz (self_ref_sim / self_canon_idx, made by make_golden.py);
* "random"  : seeded frames with repeated From strings, None matches and From == To rows.
"""
import json
import os
import sys
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.modules["seaborn"] = types.ModuleType("seaborn")
_rf = types.ModuleType("rapidfuzz")
_rf.fuzz = types.ModuleType("rapidfuzz.fuzz")
_rf.process = types.ModuleType("rapidfuzz.process")
_rf.fuzz.ratio = _rf.fuzz.WRatio = None
sys.modules.update({"rapidfuzz": _rf, "rapidfuzz.fuzz": _rf.fuzz, "rapidfuzz.process": _rf.process})

from polyfuzz.linkage import single_linkage            # noqa: E402
from polyfuzz.metrics import precision_recall_curve    # noqa: E402
from polyfuzz.models import TFIDF                      # noqa: E402


def frame_records(df):
    return {"From": df["From"].tolist(), "To": [None if (t is None or t != t) else t for t in df["To"].tolist()],
            "Similarity": [float(x) for x in df["Similarity"].tolist()]}


def linkage_records(df, thresholds):
    out = []
    for thr in thresholds:
        clusters, mapping, names = single_linkage(df, thr)
        out.append({"min_similarity": thr,
                    "clusters": [[int(k), v] for k, v in clusters.items()],
                    "cluster_mapping": [[k, int(v)] for k, v in mapping.items()],
                    "cluster_name_map": [[k, v] for k, v in names.items()]})
    return out


def pr_records(df, steps=(0.01, 0.05, 0.003)):
    out = []
    for st in steps:
        p, r, ap = precision_recall_curve(df, st)
        out.append({"precision_steps": st, "min_precisions": [float(x) for x in p], "recall": [float(x) for x in r],
                    "average_precision": [None if x != x else float(x) for x in ap]})
    return out


def main():
    cases = {}
    fl = ["apple", "apples", "appl", "recal", "house", "similarity"]
    tl = ["apple", "apples", "mouse"]
    df = TFIDF(cosine_method="sklearn", min_similarity=0).match(fl, tl)
    ths = [0, .1, .2, .3, .4, .5, .6, .7, .8, .9, 1.]
    cases["readme"] = {"frame": frame_records(df), "linkage": linkage_records(df, ths), "pr": pr_records(df)}
    dfs = TFIDF(cosine_method="sklearn", min_similarity=0).match(fl)
    cases["readme_self"] = {"frame": frame_records(dfs), "linkage": linkage_records(dfs, [0.5, 0.75]), "pr": pr_records(dfs)}

    g = np.load(os.path.join(HERE, "golden.npz"))
    sl = json.load(open(os.path.join(HERE, "company_self_list.json")))["from_list"]
    sim, idx = g["self_ref_sim"][:, 0], g["self_canon_idx"][:, 0]
    to = [None if s < 0.001 else sl[j] for s, j in zip(sim, idx)]
    dself = pd.DataFrame({"From": sl, "To": to, "Similarity": sim})
    cases["self"] = {"frame": {"Similarity": [float(x) for x in sim], "to_index": [(-1 if t is None else int(j)) for t, j in zip(to, idx)]},
                     "linkage": linkage_records(dself, [0.5, 0.75, 0.9]), "pr": pr_records(dself)}

    rng = np.random.default_rng(7)
    rnd = []
    for trial in range(12):
        n = int(rng.integers(5, 60))
        pool = [f"s{i}" for i in range(int(rng.integers(3, 30)))]
        fr = [pool[i] for i in rng.integers(0, len(pool), n)]
        to = [(None if rng.random() < 0.1 else pool[i]) for i in rng.integers(0, len(pool), n)]
        sim = [0.0 if t is None else float(s) for s, t in zip(np.round(rng.random(n), 3), to)]
        d = pd.DataFrame({"From": fr, "To": to, "Similarity": sim})
        rnd.append({"frame": frame_records(d), "linkage": linkage_records(d, [-0.5, 0.0, 0.4, 0.8]), "pr": pr_records(d, (0.01,))})
    cases["random"] = rnd
    # scores on rapidfuzz's 0..100 scale and an all-equal column
    d100 = pd.DataFrame({"From": ["a"] * 50, "To": ["b"] * 50, "Similarity": [float(x) for x in np.round(rng.random(50) * 100, 6)]})
    dsame = pd.DataFrame({"From": ["a"] * 9, "To": ["b"] * 9, "Similarity": [0.5] * 9})
    cases["scales"] = [{"frame": frame_records(d100), "pr": pr_records(d100, (0.01,))},
                       {"frame": frame_records(dsame), "pr": pr_records(dsame, (0.01, 0.5))}]
    with open(os.path.join(HERE, "group_golden.json"), "w") as f:
        json.dump(cases, f)
    print("wrote group_golden.json", os.path.getsize(os.path.join(HERE, "group_golden.json")), "bytes")


if __name__ == "__main__":
    main()
