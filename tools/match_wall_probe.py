"""Per-call wall time of `TFIDF.match(names)` in a FRESH process (is the first second slow? bimodal?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t00 = time.perf_counter()
import polyfuzz_amd
from polyfuzz_amd import datasets
from polyfuzz_amd.models import TFIDF
names = datasets.load_company_names()
ctx = polyfuzz_amd.Context.default()
print("startup %.2f s" % (time.perf_counter() - t00))
m = TFIDF(min_similarity=0, top_n=5)
keep = os.environ.get("PROBE_KEEP") == "1"
held = []
for i in range(int(os.environ.get("PROBE_N", "40"))):
    df = None                # (freeing the previous frame is not part of the call)
    t0 = time.perf_counter()
    df = m.match(names)
    dt = (time.perf_counter() - t0) * 1e3
    if keep:
        held.append(df)
    print(i, "%.2f ms" % dt, {k: round(v, 2) for k, v in m.last_timings.items()}, "t=%.2f" % (time.perf_counter() - t00))
    if os.environ.get("PROBE_SLEEP"):
        time.sleep(float(os.environ["PROBE_SLEEP"]))
