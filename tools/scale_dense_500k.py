"""BASELINE config 5 on ONE GPU's share: a 62 500-row from-shard of 768-d fp32 embeddings against 500 000
to-vectors, cosine top-10 (the 8-GPU job is 8 such shards; the to-side, 1.5 GB, is replicated).  Checks a row
sample against the float64 oracle and prints timings.  Host buffers in, host buffers out (PCIe included)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polyfuzz_amd
from polyfuzz_amd import _lib
import oracle

n_to, n_from, d, top_n = 500_000, 62_500, 768, 10
if len(sys.argv) > 1:                      # smaller run: python tools/scale_dense_500k.py <n_to> <n_from>
    n_to, n_from = int(sys.argv[1]), int(sys.argv[2])
ctx = polyfuzz_amd.Context.default()
rng = np.random.default_rng(7)
t0 = time.time()
b = rng.standard_normal((n_to, d), dtype=np.float32)
a = rng.standard_normal((n_from, d), dtype=np.float32)
# plant near-duplicates so that the top ranks are not all noise
pick = rng.choice(n_to, n_from, replace=False)
a += 2.0 * b[pick]
t_gen = time.time() - t0
_lib.dense_cossim_topn_host(ctx, a[:2048], b[:4096], top_n, 0.0)            # warm-up (allocator, code objects)
ctx.prof_enable(True); ctx.prof_reset()
t0 = time.perf_counter()
idx, val = _lib.dense_cossim_topn_host(ctx, a, b, top_n, 0.0)
dt = time.perf_counter() - t0
gemm_ms, n_panels = ctx.prof_get("k5_gemm_panel")
topn_ms, _ = ctx.prof_get("k5_row_topn")
ctx.prof_enable(False)
# the same shard with both operands resident in HBM (pipeline.DenseMatchJob: what a rank of the 8-GPU job times)
from polyfuzz_amd import pipeline
job = pipeline.DenseMatchJob(ctx, a, b, top_n=top_n)
job.step(); ctx.sync()
t0 = time.perf_counter()
for _ in range(2): r = job.step()
ctx.sync(); dt_res = (time.perf_counter() - t0) / 2
r_idx, r_val = r.download()
resident_equal = bool(np.array_equal(r_idx, idx) and np.array_equal(r_val, val))
del job
rows = rng.choice(n_from, 8, replace=False)
bad = 0; err = 0.0
for i in rows:
    e_idx, e_val = oracle.dense_cossim_topn(a[i:i + 1], b, top_n, 0.0)
    err = max(err, float(np.abs(val[i] - e_val[0]).max()))
    bad += int(not np.array_equal(idx[i], e_idx[0]))
print(json.dumps({"n_from_shard": n_from, "n_to": n_to, "dim": d, "top_n": top_n, "end_to_end_s": dt,
                  "gemm_ms": gemm_ms, "gemm_panels": n_panels, "gemm_tflops": 2.0 * n_from * n_to * d / gemm_ms / 1e9,
                  "row_topn_ms": topn_ms, "resident_step_s": dt_res,
                  "resident_step_tflops": 2.0 * n_from * n_to * d / dt_res / 1e12, "resident_equals_host_path": resident_equal,
                  "pairs_per_s_end_to_end": n_from * n_to / dt,
                  "planted_match_found_top1": float((idx[:, 0] == pick).mean()),
                  "sample_rows": len(rows), "rows_with_index_diff": bad, "max_abs_score_err": err,
                  "host_generation_s": t_gen}))
