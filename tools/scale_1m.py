"""BASELINE config 4 on ONE GPU's share: a 125k-row from-shard against 1M to-strings, top-10
(the 8-GPU job is 8 such shards).  Checks a row sample against the oracle and prints timings."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polyfuzz_amd
from polyfuzz_amd import _lib, synth, pipeline
import oracle
ctx = polyfuzz_amd.Context.default()
n_to, n_from = 1_000_000, 125_000
t0 = time.time(); tl = synth.company_names(n_to, 5678); fl = synth.company_names(n_from, 1234); t_gen = time.time() - t0
job = pipeline.TfidfMatchJob(ctx, fl, tl, top_n=10, min_similarity=0.0)
job.step(); ctx.sync()
ctx.prof_enable(True); ctx.prof_reset(); t0 = time.perf_counter()
for _ in range(2): res = job.step()
ctx.sync(); dt = (time.perf_counter() - t0) / 2
k3, n = ctx.prof_get("k3_cossim_topn")
idx, val = res.download()
a3, b3, ncol = job.host_matrices()
rows = np.random.default_rng(0).choice(n_from, 8, replace=False)
bad = 0; err = 0.0
for i in rows:
    e_idx, e_val = oracle.cossim_topn(a3, b3, ncol, 10, 0.0, rows=(int(i), int(i) + 1))
    err = max(err, float(np.abs(val[i] - e_val[0]).max()))
    bad += int(not np.array_equal(idx[i], e_idx[0]))
print(json.dumps({"n_from_shard": n_from, "n_to": n_to, "top_n": 10, "step_s": dt, "k3_ms": k3 / n, "pairs_per_s": n_from * n_to / dt,
                  "index": job.index.info(), "stats": job.stats(), "sample_rows": len(rows), "rows_with_index_diff": bad,
                  "max_abs_score_err": err, "host_generation_s": t_gen}))
