"""Where does the wall time of `.match()` go?  Stage timings of the single-query and the 100k x 100k paths."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polyfuzz_amd
from polyfuzz_amd import _lib, synth
from polyfuzz_amd.models import TFIDF
from polyfuzz_amd.models._utils import topn_to_frame
ctx = polyfuzz_amd.Context.default()
fl, tl = synth.company_names(100_000, 1234), synth.company_names(100_000, 5678)
m = TFIDF(min_similarity=0, top_n=1); m.match(fl, tl)

def t(fn, reps=20):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps * 1e3, r
q = fl[:1]
ms, s = t(lambda: _lib.DeviceStrings.upload(ctx, q)); print(f"query: pack+upload {ms:.3f} ms")
ms, a = t(lambda: m._dev_vec.transform(_lib.DeviceStrings.upload(ctx, q))); print(f"query: upload+transform {ms:.3f} ms")
ms, r = t(lambda: _lib.cossim_topn(ctx, m._dev_index, a, 1, 0.0).download()); print(f"query: K3+download {ms:.3f} ms")
ms, _ = t(lambda: topn_to_frame(r[0], r[1], q, tl, 1)); print(f"query: DataFrame {ms:.3f} ms")
ms, _ = t(lambda: m.match(q, tl, re_train=False)); print(f"query: match() total {ms:.3f} ms")
ms, s = t(lambda: _lib.pack_strings(fl), 3); print(f"100k: pack_strings {ms:.2f} ms")
ms, s = t(lambda: _lib.DeviceStrings.upload(ctx, fl), 3); print(f"100k: pack+upload {ms:.2f} ms")
m5 = TFIDF(min_similarity=0, top_n=5)
ms, df = t(lambda: m5.match(fl, tl), 3); print(f"100k: match() total {ms:.2f} ms")
idx = np.random.default_rng(0).integers(0, 100000, (100000, 5)).astype(np.int32); val = np.random.default_rng(1).random((100000, 5)).astype(np.float32)
ms, _ = t(lambda: topn_to_frame(idx, val, fl, tl, 5), 3); print(f"100k: DataFrame(top5) {ms:.2f} ms")
