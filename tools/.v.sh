timeout 300 python -m pytest tests/test_dense_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
python - <<'PY'
import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import polyfuzz_amd
from polyfuzz_amd import _lib
ctx = polyfuzz_amd.Context.default()
rng = np.random.default_rng(0)
for n, d in ((20000, 768), (50000, 768), (20000, 384)):
    a = rng.standard_normal((n, d)).astype(np.float32); b = rng.standard_normal((n, d)).astype(np.float32)
    for env in ("", "1"):
        if env: os.environ["PFZ_K5_NO_PIPE"] = "1"
        else: os.environ.pop("PFZ_K5_NO_PIPE", None)
        _lib.dense_cossim_topn_host(ctx, a[:4096], b[:4096], 10, 0.0)
        ctx.prof_enable(True); ctx.prof_reset()
        t = time.perf_counter(); idx, val = _lib.dense_cossim_topn_host(ctx, a, b, 10, 0.0); dt = time.perf_counter() - t
        ms, cnt = ctx.prof_get("k5_gemm_panel"); ctx.prof_enable(False)
        print(f"n={n} d={d} pipe={'off' if env else 'on '}: gemm {ms:.2f} ms = {2*n*n*d/ms/1e9:.1f} TFLOP/s, end to end {dt*1e3:.1f} ms", flush=True)
PY
