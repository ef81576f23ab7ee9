"""TFIDF.match(names) wall time and K3 device time for several ways of cutting the K3 launch into parts (the frame columns
of a part are built while the device works on the next ones).  usage: python tools/match_split_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polyfuzz_amd
from polyfuzz_amd import datasets
from polyfuzz_amd.models import TFIDF, _tfidf
names = datasets.load_company_names()
ctx = polyfuzz_amd.Context.default()
SHARES = [(1.0,), (0.3, 0.3, 0.25, 0.15), (0.1,) + (0.15,) * 6, (0.08,) + (0.115,) * 8, (0.06, 0.1, 0.12, 0.14, 0.14, 0.14, 0.15, 0.15),
          (0.1, 0.2, 0.2, 0.2, 0.15, 0.15), (0.15, 0.25, 0.25, 0.2, 0.15), (0.05,) + (0.095,) * 10, (0.1, 0.15, 0.15, 0.15, 0.15, 0.1, 0.1, 0.1)]
if len(sys.argv) > 1:      # python tools/match_split_probe.py 0.4,0.3,0.2,0.1 0.5,0.5 ...
    SHARES = [tuple(float(x) for x in a.split(",")) for a in sys.argv[1:]]
for shares in SHARES:
    os.environ["PFZ_MATCH_SHARES"] = ",".join(str(x) for x in shares)      # (read by _tfidf._split_ends on every match)
    m = TFIDF(min_similarity=0, top_n=5)
    for _ in range(3):
        m.match(names)
    ts, st = [], []
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(9):
        df = None
        t0 = time.perf_counter(); df = m.match(names); ts.append((time.perf_counter() - t0) * 1e3); st.append(m.last_timings)
    k3, n = ctx.prof_get("k3_cossim_topn"); ctx.prof_enable(False)
    i = int(np.argsort(ts)[len(ts) // 2])
    print(f"shares {shares}: wall median {ts[i]:.2f} ms (min {min(ts):.2f}); K3 {k3 / 9:.2f} ms in {n // 9} launches; stages "
          + ", ".join(f"{k} {v:.2f}" for k, v in st[i].items()), flush=True)
