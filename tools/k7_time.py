"""K7 on config 3's lists: RapidFuzz default scorer (WRatio) and its parts, n_from x 20 000 IMDB titles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polyfuzz_amd
from polyfuzz_amd import _lib, datasets
ctx = polyfuzz_amd.Context.default()
fl, tl = datasets.c3_lists()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["WRatio"]
fl = [s for s in fl if len(s) <= 128 and len(set(s.split())) <= 32][:n]
for mode in modes:
    _lib.fuzz_extract_one(ctx, fl[:64], tl, mode)
    ctx.prof_enable(True); ctx.prof_reset()
    t0 = time.perf_counter()
    idx, score = _lib.fuzz_extract_one(ctx, fl, tl, mode)
    wall = time.perf_counter() - t0
    k_ms, k_n = ctx.prof_get("k7_fuzz")
    ctx.prof_enable(False)
    print(f"{mode:26s} {len(fl)} x {len(tl)}: wall {wall * 1e3:9.1f} ms, kernels {k_ms:9.1f} ms ({k_n} scopes) = "
          f"{len(fl) * len(tl) / max(k_ms, 1e-9) / 1e6:8.2f} G pairs/s; mean score {score.mean():.3f}")
