"""Scratch micro-benchmark of K3 alone on CSR inputs vectorised on the host (sklearn).
Not the judged bench (bench.py) -- used while tuning the kernel."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polyfuzz_amd
from polyfuzz_amd import _lib, synth

def vectorize(from_list, to_list):
    import re
    from sklearn.feature_extraction.text import TfidfVectorizer
    def ana(s):
        s = re.sub(r'[^A-Za-z0-9 ]+', '', s.lower()); s = re.sub(r'\s+', ' ', s).strip()
        return [s[i:i+3] for i in range(len(s)-2) if ' ' not in s[i:i+3]]
    v = TfidfVectorizer(min_df=1, analyzer=ana).fit(to_list + from_list)
    return v.transform(from_list), v.transform(to_list)

n_from = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_to = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
ntop = int(sys.argv[3]) if len(sys.argv) > 3 else 5
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
t = time.time()
A, B = vectorize(synth.company_names(n_from, 1234), synth.company_names(n_to, 5678))
print("vectorize host s", time.time() - t, A.shape, A.nnz, B.nnz, flush=True)
dfa = np.bincount(A.indices, minlength=A.shape[1]).astype(np.float64)
dfb = np.bincount(B.indices, minlength=A.shape[1]).astype(np.float64)
madds = float((dfa * dfb).sum())
ctx = polyfuzz_amd.Context(0)
print(ctx.info())
dA = _lib.DeviceCSR.from_scipy(ctx, A); dB = _lib.DeviceCSR.from_scipy(ctx, B)
ctx.prof_enable(True)
ix = _lib.DeviceIndex.build(ctx, dB)
out = _lib.DeviceTopN.alloc(ctx, n_from, ntop)
_lib.cossim_topn(ctx, ix, dA, ntop, 0.0, out=out); ctx.sync()
ctx.prof_reset()
for _ in range(steps):
    ix2 = _lib.DeviceIndex.build(ctx, dB)
    _lib.cossim_topn(ctx, ix2, dA, ntop, 0.0, out=out)
ctx.sync()
res = {}
for k in ("k_index_count", "k_index_fill", "k3_cossim_topn"):
    ms, n = ctx.prof_get(k); res[k] = ms / max(n, 1)
k3 = res["k3_cossim_topn"] * 1e-3
bytes_alg = 8 * madds + 8 * A.nnz + 8 * n_from * ntop
print(json.dumps({"ms": res, "madds": madds, "pairs_per_s": n_from * n_to / k3, "alg_GBs": bytes_alg / k3 / 1e9,
                  "madds_per_s": madds / k3, "index": ix.info()}))
