"""GEMM time of K5 on n x n x 768 (default 32768): random-normal, small-integer and all-zero fills -- the MFMA rate
on this part depends on the operand data (power), so a roofline fraction should say which fill it was measured on."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, polyfuzz_amd
from polyfuzz_amd import pipeline
ctx = polyfuzz_amd.Context.default()
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
fills = {"random normal": lambda: rng.standard_normal((n, 768), dtype=np.float32),
         "small integers": lambda: rng.integers(-3, 4, (n, 768)).astype(np.float32),
         "zeros": lambda: np.zeros((n, 768), np.float32)}
import os
for name, mk in list(fills.items())[:int(os.environ.get('K5_FILLS', '3'))]:
    a = mk()
    job = pipeline.DenseMatchJob(ctx, a, a, top_n=10, normalize=False)
    job.step(); ctx.sync()
    ts = []
    for _ in range(6):
        ctx.prof_enable(2); ctx.prof_reset(); job.step(); ctx.sync(); ts.append(ctx.prof_get("k5_gemm_panel")[0])
    print(f"{name:15s} GEMM ms {min(ts):8.3f}  {2.0 * n * n * 768 / min(ts) / 1e9:6.1f} TFLOP/s")
    del job
