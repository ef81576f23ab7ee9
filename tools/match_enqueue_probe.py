"""Where the host time of a TFIDF.match() goes before the device has everything it needs: pack, upload, fit, transform, index, the K3 launches (median of 15).
usage (GPU box): python tools/match_enqueue_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np
from polyfuzz_amd import datasets, _lib
names = datasets.load_company_names()
ctx = _lib.Context.default()
P = _lib.TfidfParams(3, 3, 1, 1)
def once():
    t = [time.perf_counter()]
    col = np.empty(len(names), dtype=object)
    packed = _lib.pack_strings(names, col); t.append(time.perf_counter())
    s = _lib.DeviceStrings.upload_packed(ctx, *packed); t.append(time.perf_counter())
    vec = _lib.DeviceTfidf.fit(ctx, P, s, None); t.append(time.perf_counter())
    a = vec.transform(s); t.append(time.perf_counter())
    ix = _lib.DeviceIndex.build(ctx, a); t.append(time.perf_counter())
    res = None
    for lo, hi in ((0, 30000), (30000, 60000), (60000, 85000), (85000, 100000)):
        res = _lib.cossim_topn(ctx, ix, a, 5, 0.0, exclude_diag=True, out=res, rows=(lo, hi))
    t.append(time.perf_counter())
    ctx.sync(); t.append(time.perf_counter())
    return np.diff(t) * 1e3
for _ in range(5): once()
r = np.median([once() for _ in range(15)], axis=0)
print('pack %.3f  upload %.3f  fit %.3f  transform %.3f  index %.3f  k3 enqueue %.3f  drain %.3f' % tuple(r))
