"""Pin script for the third-party call at polyfuzz/models/_utils.py:9,82:
`sparse_dot_topn.awesome_cossim_topn(from_vector, to_vector.T, top_n + 1, min_similarity)` -- per row the `ntop` largest
products that are STRICTLY above `lower_bound`, as a CSR matrix.

sparse_dot_topn (setup.py:28, `sparse_dot_topn>=0.2.9`) is NOT installable in the build container (no wheel, no network), so
oracle/cossim_topn.c restates those two rules from the library's published behaviour and is pinned only on the reference's
sklearn back-end (which ignores min_similarity).  This script closes that wherever the library is at hand:

    python tests/golden/pin_sparse_dot_topn.py           # importable: regenerate from the REAL library, diff vs the oracle
    python tests/golden/pin_sparse_dot_topn.py --oracle  # not importable: (re)write the fixture from the oracle

It writes tests/golden/sparse_dot_topn_pin.json: for the README lists and for 300 x 291 real company names (the golden C2
lists' first rows, vectorised by scikit-learn exactly as _tfidf.py:102-118 does), at (ntop, lower_bound) = (2, 0.75) -- what
TFIDF() calls by default --, (6, 0.0), (4, 0.3) and a lower_bound EQUAL to an occurring score (the strict `>`): per row the
kept columns in (score desc, column asc) order and their scores.  tests/test_oracle_cpu.py holds oracle/cossim_topn.c to it.
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def _analyzer(s):
    s = re.sub(r'[^A-Za-z0-9 ]+', '', s.lower())
    s = re.sub(r'\s+', ' ', s).strip()
    return [s[i:i + 3] for i in range(len(s) - 2) if ' ' not in s[i:i + 3]]


def matrices(from_list, to_list):
    from sklearn.feature_extraction.text import TfidfVectorizer
    v = TfidfVectorizer(min_df=1, analyzer=_analyzer).fit(to_list + from_list)      # _tfidf.py:109-110
    a, b = v.transform(from_list).tocsr(), v.transform(to_list).tocsr()
    a.sort_indices()
    b.sort_indices()
    return a, b


def cases():
    readme = (["apple", "apples", "appl", "recal", "house", "similarity"], ["apple", "apples", "mouse"])
    c2 = json.load(open(os.path.join(HERE, "company_c2_lists.json"), encoding="utf-8"))
    lists = {"readme": readme, "companies": (c2["from_list"][:300], c2["to_list"][:291])}
    out = []
    for name, (fl, tl) in lists.items():
        a, b = matrices(fl, tl)
        dense = (a @ b.T).toarray()
        exact = float(np.sort(dense[dense > 0.2].ravel())[len(dense[dense > 0.2]) // 2]) if (dense > 0.2).any() else 0.5
        for ntop, lb in ((2, 0.75), (6, 0.0), (4, 0.3), (3, exact)):
            out.append((name, fl, tl, a, b, ntop, lb))
    return out


def canon(csr_like_rows):
    return [[[int(j), float(v)] for v, j in sorted(((v, j) for j, v in row), key=lambda t: (-t[0], t[1]))] for row in csr_like_rows]


def from_oracle(a, b, ntop, lb):
    import oracle
    oracle.build_native()
    t = lambda m: (m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float64))
    idx, val = oracle.cossim_topn(t(a), t(b), a.shape[1], ntop, lb)
    return canon([[(int(j), float(v)) for j, v in zip(ri, rv) if j >= 0] for ri, rv in zip(idx, val)])


def from_library(a, b, ntop, lb):
    from sparse_dot_topn import awesome_cossim_topn
    m = awesome_cossim_topn(a, b.T.tocsr(), ntop, lb).tocsr()
    return canon([list(zip(m.indices[m.indptr[i]:m.indptr[i + 1]].tolist(), m.data[m.indptr[i]:m.indptr[i + 1]].tolist()))
                  for i in range(m.shape[0])])


def main(argv):
    try:
        import sparse_dot_topn  # noqa: F401
        have = "--oracle" not in argv
    except ImportError:
        have = False
    rec, bad = [], 0
    for name, fl, tl, a, b, ntop, lb in cases():
        orc = from_oracle(a, b, ntop, lb)
        rows = orc
        if have:
            rows = from_library(a, b, ntop, lb)
            for i, (x, y) in enumerate(zip(rows, orc)):
                if [c for c, _ in x] != [c for c, _ in y] or any(abs(u[1] - w[1]) > 1e-12 for u, w in zip(x, y)):
                    bad += 1
                    if bad <= 20:
                        print(f"  {name} ntop {ntop} lb {lb}: row {i}: library {x} oracle {y}")
        rec.append({"lists": name, "n_from": len(fl), "n_to": len(tl), "ntop": ntop, "lower_bound": lb, "rows": rows})
    src = "sparse_dot_topn (the library)" if have else "oracle (oracle/cossim_topn.c; PARITY UNPINNED for this third-party call until regenerated with sparse_dot_topn importable)"
    if have:
        print(f"sparse_dot_topn vs oracle/cossim_topn.c: {bad} rows differ")
    else:
        print("sparse_dot_topn is not importable here: writing the ORACLE's answers (source says so)")
    path = os.path.join(HERE, "sparse_dot_topn_pin.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump({"made_by": "tests/golden/pin_sparse_dot_topn.py", "source": src, "cases": rec}, f, separators=(",", ":"))
    print(f"{len(rec)} cases -> {path} ({os.path.getsize(path)} bytes)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
