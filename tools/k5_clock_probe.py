"""Is K5 limited by the matrix pipe or by the clock it is allowed to run at?  The same GEMM on small inputs,
cold (after a pause) and back to back, and the SMI clock while a long one runs."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polyfuzz_amd
from polyfuzz_amd import pipeline
ctx = polyfuzz_amd.Context.default()
rng = np.random.default_rng(0)
for n in (8192, 16384, 32768):
    a = rng.standard_normal((n, 768), dtype=np.float32)
    job = pipeline.DenseMatchJob(ctx, a, a, top_n=10)
    job.step(); ctx.sync()
    flop = 2.0 * n * n * 768
    for label, pause, reps in (("cold (2 s pause before each)", 2.0, 3), ("back to back", 0.0, 10)):
        ts = []
        for _ in range(reps):
            if pause:
                time.sleep(pause)
            ctx.prof_enable(2); ctx.prof_reset()
            job.step(); ctx.sync()
            ts.append(ctx.prof_get("k5_gemm_panel")[0])
        print(f"n={n:6d} {label:32s} GEMM ms {min(ts):8.3f} .. {max(ts):8.3f}   best {flop / min(ts) / 1e9:6.1f} TFLOP/s, last {flop / ts[-1] / 1e9:6.1f}")
    del job
a = rng.standard_normal((65536, 768), dtype=np.float32)
job = pipeline.DenseMatchJob(ctx, a, a, top_n=10)
job.step(); ctx.sync()
p = subprocess.Popen("for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>/dev/null | grep -i -E 'sclk|mclk' | head -2; rocm-smi --showpower 2>/dev/null | grep -i -E 'power' | head -1; sleep 0.15; done", shell=True,
                     stdout=subprocess.PIPE, text=True)
t0 = time.perf_counter()
for _ in range(4):
    job.step()
ctx.sync()
dt = (time.perf_counter() - t0) / 4
print("65536^2 x 768 step %.3f s = %.1f TFLOP/s" % (dt, 2.0 * 65536 * 65536 * 768 / dt / 1e12))
print(p.communicate()[0])
