"""Host-side cost of getting 100 000 names ready for the upload: the packer's one walk (with / without the From column) against
its threaded two-pass path, and the From column on its own (medians of 21, ms)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polyfuzz_amd import _lib, datasets
from polyfuzz_amd.models import _utils as u
names = datasets.load_company_names()
n = len(names)
def t(fn, N=21):
    fn(); ts = []
    for _ in range(N):
        t0 = time.perf_counter(); r = fn(); ts.append((time.perf_counter() - t0) * 1e3); del r
    return sorted(ts)[N // 2]
P = _lib._pack
print("one walk + From column  %.3f" % t(lambda: P.pack(names, 1, np.empty(n, dtype=object).ctypes.data) and None))
def with_col():
    c = np.empty(n, dtype=object); r = P.pack(names, 1, c.ctypes.data); return (r, c)
print("one walk + From column (array kept) %.3f" % t(with_col))
print("one walk                %.3f" % t(lambda: P.pack(names, 1)))
for k in (1, 2, 4, 8):
    print("two passes, %d thread(s)  %.3f" % (k, t(lambda: P.pack(names, -k))))
# (round 5, measured and dropped: the From column's slots filled by the packing threads with atomic reference counts -- 0.74 ms
# on four threads, the locked adds wait for every cold line; the column made by the calling thread after four threads packed
# -- 0.74 ms instead of 0.14, every line comes back from another core's cache; either way TFIDF.match(names) got SLOWER than with
# the one walk, 4.7 / 5.6 ms against 4.3 on one box, tools/r5_match.sh: the frame's gathers pay for the scattered lines too)
print("From column alone       %.3f" % t(lambda: u.object_column(names)))
print("tuple(names)            %.3f" % t(lambda: tuple(names)))
print("np.empty(n, object)     %.3f" % t(lambda: np.empty(n, dtype=object)))
