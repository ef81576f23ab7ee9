"""K7 on config 3's lists: RapidFuzz's default scorer (WRatio) and its parts, n_from x 20 000 IMDB titles, lists resident;
optionally the self-match of all 100 000 company names.  usage: python tools/k7_time.py [n_from] [modes,comma] [names]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polyfuzz_amd
from polyfuzz_amd import _lib, datasets
ctx = polyfuzz_amd.Context.default()
fl, tl = datasets.c3_lists()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["WRatio"]
fl = fl[:n]
f_dev, t_dev = _lib.DeviceStrings.upload(ctx, fl), _lib.DeviceStrings.upload(ctx, tl)
out = _lib.DeviceTopN.alloc(ctx, len(fl), 2)
t0 = time.perf_counter(); info = _lib.fuzz_plan_info(ctx, t_dev); ctx.sync()
print("plan", info, f"built in {(time.perf_counter() - t0) * 1e3:.2f} ms")
for mode in modes:
    _lib.fuzz_extract_one_dev(ctx, f_dev, t_dev, mode, out); ctx.sync()
    ctx.prof_enable(True); ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        _lib.fuzz_extract_one_dev(ctx, f_dev, t_dev, mode, out)
    ctx.sync()
    wall = (time.perf_counter() - t0) / 3
    k_ms, k_n = ctx.prof_get("k7_fuzz"); p_ms, _ = ctx.prof_get("k7_prepare")
    ctx.prof_enable(False)
    w = _lib.fuzz_extract_one_dev(ctx, f_dev, t_dev, mode, out, counters=True)
    idx, score = _lib.best_from_topn(*out.download())
    pairs = len(fl) * len(tl)
    print(f"{mode:26s} {len(fl)} x {len(tl)}: step {wall * 1e3:8.2f} ms, k7_fuzz {k_ms / 3:8.2f} ms, k7_prepare {p_ms / 3:6.3f} ms = "
          f"{pairs / max(k_ms / 3, 1e-9) / 1e6:8.2f} G pairs/s; scored {w['pairs_scored'] / pairs:.4f} of the pairs, "
          f"{w['word_steps_scored'] / max(w['pairs_scored'], 1):.0f} word-steps each; mean score {score.mean():.3f}")
if len(sys.argv) > 3:
    from polyfuzz_amd.models import RapidFuzz
    names = datasets.load_company_names()
    m = RapidFuzz()
    for rep in range(2):
        ctx.prof_enable(True); ctx.prof_reset()
        t0 = time.perf_counter(); df = m.match(names); wall = time.perf_counter() - t0
        k_ms, _ = ctx.prof_get("k7_fuzz"); p_ms, _ = ctx.prof_get("k7_prepare"); ctx.prof_enable(False)
        print(f"RapidFuzz() [WRatio] self-match of all {len(names)} company names: wall {wall:.3f} s, k7_fuzz {k_ms / 1e3:.3f} s, "
              f"k7_prepare {p_ms:.1f} ms; mean similarity {df['Similarity'].mean():.4f}")
