import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import polyfuzz_amd
from polyfuzz_amd import _lib
from tests.helpers import random_csr
ctx = polyfuzz_amd.Context.default()
rng = np.random.default_rng(5)
n_col = 600
a3 = random_csr(rng, 2000, n_col, 0.022); b3 = random_csr(rng, 6000, n_col, 0.02)
os.environ["PFZ_K3_SLICES"] = "1"
def run(**env):
    for k, v in env.items(): os.environ[k] = v
    return _lib.cossim_topn_host(ctx, a3, b3, n_col, 5, 0.0, False)
ri, rv = run(PFZ_K3_PAIR="0")
for single in ("2", "1", "0"):
    i, v = run(PFZ_K3_PAIR="1", PFZ_K3_PAIR_SINGLE=single, PFZ_K3_PAIR_DEBUG="1")
    bad = (i != ri).any(axis=1) | (v != rv).any(axis=1)
    print({"2": "single, plain scatter", "1": "single", "0": "pairs"}[single], "bad rows", int(bad.sum()), "even", int(bad[0::2].sum()), "odd", int(bad[1::2].sum()))
    for r in np.nonzero(bad)[0][:4]:
        print("  row", r, "nnz", a3[0][r+1]-a3[0][r], "got", i[r], np.round(v[r], 4), "exp", ri[r], np.round(rv[r], 4))
