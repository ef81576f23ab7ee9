"""where does the frame fill's time go on this box?  fill 100k x 5 from (a) ordinary memory, (b) the context's pinned mirror"""
import ctypes, sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import polyfuzz_amd
from polyfuzz_amd import _lib, datasets
from polyfuzz_amd.models import _utils
from polyfuzz_amd.models._tfidf import _SPLIT_EVENT, _split_ends
ctx = polyfuzz_amd.Context.default()
names = datasets.load_company_names()
n, ntop = len(names), 5
s = _lib.DeviceStrings.upload(ctx, names)
a = _lib.DeviceTfidf.fit(ctx, _lib.TfidfParams(3, 3, 1, 1), s, None).transform(s)
ends = _split_ends(n, True)
def run(copy_first):
    ts = []
    for rep in range(12):
        ix = _lib.DeviceIndex.build(ctx, a)
        res, h_idx, h_val = _lib.cossim_topn_ranges(ctx, ix, a, ntop, 0.0, True, ends, _SPLIT_EVENT, mirror=True)
        ctx.event_wait(_SPLIT_EVENT + len(ends) - 1)          # everything is there: time the fills alone
        ctx.sync()
        fb = _utils.FrameBuilder(names, names, ntop, np.empty(n, dtype=object))
        if copy_first:
            ci = np.ctypeslib.as_array(ctypes.cast(h_idx, ctypes.POINTER(ctypes.c_int32)), (n * ntop,)).copy()
            cv = np.ctypeslib.as_array(ctypes.cast(h_val, ctypes.POINTER(ctypes.c_float)), (n * ntop,)).copy()
            bi, bv = ci.ctypes.data, cv.ctypes.data
        else:
            bi, bv = h_idx, h_val
        t0 = time.perf_counter()
        row0 = 0
        for row1 in ends:
            fb.fill_raw(bi + 4 * ntop * row0, bv + 4 * ntop * row0, row1 - row0, row0)
            row0 = row1
        ts.append((time.perf_counter() - t0) * 1e3)
        fb.from_col[:] = None
        del fb
    ts.sort()
    return ts[len(ts) // 2]
for k in range(2):
    print("fill of 100k x 5 from the pinned mirror: %.3f ms   from an ordinary copy: %.3f ms" % (run(False), run(True)))
