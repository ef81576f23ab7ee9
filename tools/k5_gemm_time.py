import sys, time; sys.path.insert(0,'.')
import numpy as np, polyfuzz_amd
from polyfuzz_amd import pipeline
ctx = polyfuzz_amd.Context.default()
rng = np.random.default_rng(0)
n = 32768
a = rng.standard_normal((n, 768), dtype=np.float32)
job = pipeline.DenseMatchJob(ctx, a, a, top_n=10)
job.step(); ctx.sync()
ts = []
for _ in range(6):
    ctx.prof_enable(2); ctx.prof_reset(); job.step(); ctx.sync(); ts.append(ctx.prof_get("k5_gemm_panel")[0])
print("GEMM ms", min(ts), "TFLOP/s", 2.0 * n * n * 768 / min(ts) / 1e9)
