"""Experiment: does the ORDER of the from-rows matter for K3 (L2 locality of the heavy posting lists)?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polyfuzz_amd
from polyfuzz_amd import _lib, synth, pipeline
ctx = polyfuzz_amd.Context.default()
tl = synth.company_names(100_000, 5678)
fl = synth.company_names(100_000, 1234)
orders = {"random (as generated)": fl,
          "sorted by last token": sorted(fl, key=lambda s: (s.split() or [""])[-1]),
          "sorted alphabetically": sorted(fl)}
for name, lst in orders.items():
    job = pipeline.TfidfMatchJob(ctx, lst, tl, top_n=5, min_similarity=0.0)
    for _ in range(2): job.step()
    ctx.sync(); ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(5): job.step()
    ctx.sync(); ms, n = ctx.prof_get("k3_cossim_topn"); ctx.prof_enable(False)
    print(f"{name:28s} k3 {ms / n:.3f} ms")
