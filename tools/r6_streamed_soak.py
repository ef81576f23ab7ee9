"""round 6: a soak of the streamed self-match (k3_sym_launch_streamed) -- its hand-over between a running pass-1 kernel and the
merges on a side stream is a matter of memory ordering (write-through stores, counters, a kernel start's cache invalidation, words
in pinned host memory), and a race would not show on every run.  N matches of the headline list, every one consumed the way
TFIDF.match consumes it (the host mirror, range by range, the moment its word arrives) and compared bit for bit with the row-major
kernel's result; the range cuts vary from run to run.  usage: python tools/r6_streamed_soak.py [runs=300]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polyfuzz_amd
from polyfuzz_amd import _lib, datasets
from polyfuzz_amd.models._tfidf import _SPLIT_EVENT

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = polyfuzz_amd.Context.default()
names = datasets.load_company_names()
n, ntop = len(names), 5
s = _lib.DeviceStrings.upload(ctx, names)
a = _lib.DeviceTfidf.fit(ctx, _lib.TfidfParams(3, 3, 1, 1), s, None).transform(s)
os.environ["PFZ_K3_SYM"] = "0"
r_idx, r_val = _lib.cossim_topn(ctx, _lib.DeviceIndex.build(ctx, a), a, ntop, 0.0, exclude_diag=True).download()
del os.environ["PFZ_K3_SYM"]
rng = np.random.default_rng(6)


def at(addr, ctype, dtype, r0, r1):
    m = (r1 - r0) * ntop
    return np.ctypeslib.as_array(ctypes.cast(addr, ctypes.POINTER(ctype)), (m,)).reshape(-1, ntop).astype(dtype)


bad = 0
t0 = time.time()
for run in range(runs):
    k = int(rng.integers(1, 17))
    cuts = sorted(set((rng.choice(np.arange(1, n // 2048 + 1), size=k - 1, replace=False) * 2048).tolist())) if k > 1 else []
    ends = [c for c in cuts if c < n] + [n]
    ix = _lib.DeviceIndex.build(ctx, a)
    res, h_idx, h_val = _lib.cossim_topn_ranges(ctx, ix, a, ntop, 0.0, True, ends, _SPLIT_EVENT, mirror=True)
    assert h_idx, "the streamed form did not run"
    row0 = 0
    for i, row1 in enumerate(ends):
        ctx.event_wait(_SPLIT_EVENT + i)
        g_idx = at(h_idx + 4 * ntop * row0, ctypes.c_int32, np.int32, row0, row1)
        g_val = at(h_val + 4 * ntop * row0, ctypes.c_float, np.float32, row0, row1)
        if not (np.array_equal(g_idx, r_idx[row0:row1]) and np.array_equal(g_val, r_val[row0:row1])):
            bad += 1
            d = np.flatnonzero((g_idx != r_idx[row0:row1]).any(axis=1) | (g_val != r_val[row0:row1]).any(axis=1))
            print(f"run {run}: range {i} [{row0}, {row1}) of {ends}: {len(d)} rows differ in the host mirror, first {row0 + d[:5]}", flush=True)
        row0 = row1
    d_idx, d_val = res.download()
    if not (np.array_equal(d_idx, r_idx) and np.array_equal(d_val, r_val)):
        bad += 1
        print(f"run {run}: the device result differs", flush=True)
print(f"{runs} streamed self-matches of {n} names (1 - 16 ranges each, random cuts), {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
