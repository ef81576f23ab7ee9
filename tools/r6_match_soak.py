"""round 6: soak of the user-level call with its host threads -- TFIDF(min_similarity=0, top_n=5).match(names) N times in a row
(the string packer and the frame's range fill on the pool's crews, the ranges' chains on two side streams): every frame equal to the
first cell for cell (pointers and similarities), the names' reference counts where they were.   usage: python tools/r6_match_soak.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polyfuzz_amd import datasets
from polyfuzz_amd.models import TFIDF
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
names = datasets.load_company_names()
m = TFIDF(min_similarity=0, top_n=5)
first = m.match(names)
cols = list(first.columns)
ref_ids = {c: np.fromiter((id(x) for x in first[c].to_numpy()), np.int64, len(first)) for c in cols if not c.startswith("Similarity")}
ref_sim = {c: first[c].to_numpy().copy() for c in cols if c.startswith("Similarity")}
rc0 = [sys.getrefcount(s) for s in names[::97]]
bad = 0
t0 = time.perf_counter()
for run in range(n_runs):
    df = m.match(names)
    ok = list(df.columns) == cols
    for c in cols:
        a = df[c].to_numpy()
        if c.startswith("Similarity"):
            ok = ok and np.array_equal(a, ref_sim[c])
        elif run % 50 == 0:            # (the pointer columns of every fiftieth frame: 600 000 ids each)
            ok = ok and np.array_equal(np.fromiter((id(x) for x in a), np.int64, len(a)), ref_ids[c])
    bad += not ok
    del df, a
dt = time.perf_counter() - t0
rc1 = [sys.getrefcount(s) for s in names[::97]]
print(f"{n_runs} calls of TFIDF.match({len(names)} names), {bad} frames that differ from the first, reference counts "
      f"{'unchanged' if rc1 == rc0 else 'CHANGED'}, {dt:.1f} s")
