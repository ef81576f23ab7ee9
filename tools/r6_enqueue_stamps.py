import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from polyfuzz_amd import datasets, _lib
from polyfuzz_amd.models import TFIDF, _utils, _tfidf
names = datasets.load_company_names()
ctx = _lib.Context.default()
m = TFIDF(min_similarity=0, top_n=5)
for _ in range(5): m.match(names)
# the stages of the enqueue, by hand (same calls as TFIDF.match), median of 15
rows = []
for rep in range(15):
    t = [time.perf_counter()]
    s = _lib.DeviceStrings.upload_direct(ctx, names, None); t.append(time.perf_counter())
    vec = _lib.DeviceTfidf.fit(ctx, m._params(), s, None); t.append(time.perf_counter())
    a = vec.transform(s); t.append(time.perf_counter())
    ix = _lib.DeviceIndex.build(ctx, a); t.append(time.perf_counter())
    ends = _tfidf._split_ends(len(names), True)
    res, hi, hv = _lib.cossim_topn_ranges(ctx, ix, a, 5, 0.0, True, ends, _tfidf._SPLIT_EVENT, mirror=True); t.append(time.perf_counter())
    ctx.event_wait(_tfidf._SPLIT_EVENT + len(ends) - 1); t.append(time.perf_counter())
    ctx.sync()
    rows.append(np.diff(t) * 1e3)
r = np.median(np.array(rows), axis=0)
print("pack+upload %.3f  fit %.3f  transform %.3f  index %.3f  ranges enqueue %.3f  wait for last range %.3f   (sum %.3f)" % (*r, r.sum()))
