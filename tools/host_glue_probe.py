"""Host-side glue of TFIDF.match on this box: string packing and frame columns vs helper thread counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polyfuzz_amd import _lib, datasets
from polyfuzz_amd.models import _utils
names = datasets.load_company_names()
rng = np.random.default_rng(0)
idx = np.clip(np.arange(100000)[:, None] + rng.integers(-3000, 3000, (100000, 5)), 0, 99999).astype(np.int32)
val = rng.random((100000, 5)).astype(np.float32)


def med(f, n=15):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


for th in (1, 2, 4, 8, 16):
    p = med(lambda: _lib._pack.pack(names, th))
    _utils._RANGE_THREADS = th
    fr = med(lambda: _utils.topn_to_frame(idx, val, names, names, 5))
    print(f"threads {th:2d}: pack {p:.3f} ms   frame(top-5, 100k) {fr:.3f} ms")
print("from column", med(lambda: _utils.object_column(names)))
