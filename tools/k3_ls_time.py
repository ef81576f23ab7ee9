"""K3 on one GPU's shard of config 4 (125 000 x 1 000 000 synthetic names, top-10): the row-major kernel against the
lock-step kernel (k3_lockstep.hip) at several slice widths / block sizes; every variant's result must equal the first's
bit for bit.  usage: python tools/k3_ls_time.py [n_to] [n_from] [variants...]   variant = BLOCK:LOCKSTEP:S[:WAVES[:CHUNK]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polyfuzz_amd
from polyfuzz_amd import _lib, synth
n_to = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_from = int(sys.argv[2]) if len(sys.argv) > 2 else 125_000
variants = sys.argv[3:] or ["4096:0:1", "4096:1:4", "2048:1:8"]
ctx = polyfuzz_amd.Context.default()
tl, fl = synth.company_names(n_to, 5678), synth.company_names(n_from, 1234)
t = _lib.DeviceStrings.upload(ctx, tl); f = _lib.DeviceStrings.upload(ctx, fl)
vec = _lib.DeviceTfidf.fit(ctx, _lib.TfidfParams(3, 3, 1, 1), t, f)
a, b = vec.transform(f), vec.transform(t)
out = _lib.DeviceTopN.alloc(ctx, n_from, 10)
ref = None
for v in variants:
    p = v.split(":")
    os.environ["PFZ_K3_BLOCK"], os.environ["PFZ_K3_LOCKSTEP"], os.environ["PFZ_K3_LS_BLOCKS"] = p[0], p[1], p[2]
    if len(p) > 3 and p[3]: os.environ["PFZ_K3_LS_WAVES"] = p[3]
    else: os.environ.pop("PFZ_K3_LS_WAVES", None)
    if len(p) > 4: os.environ["PFZ_K3_LS_CHUNK"] = p[4]
    else: os.environ.pop("PFZ_K3_LS_CHUNK", None)
    ix = _lib.DeviceIndex.build(ctx, b)
    _lib.cossim_topn(ctx, ix, a, 10, 0.0, out=out); ctx.sync()
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(3):
        _lib.cossim_topn(ctx, ix, a, 10, 0.0, out=out)
    ctx.sync()
    ms, n = ctx.prof_get("k3_cossim_topn"); ctx.prof_enable(False)
    idx, val = out.download()
    same = "first" if ref is None else (bool(np.array_equal(idx, ref[0]) and np.array_equal(val, ref[1])))
    if ref is None: ref = (idx.copy(), val.copy())
    print(f"block {p[0]} lockstep {p[1]} S {p[2]} waves {p[3] if len(p) > 3 and p[3] else 'auto'} chunk {p[4] if len(p) > 4 else 'auto'}: k3 {ms / n:8.2f} ms   index {ix.info()['n_pieces'] * 128 / 1e6:.0f} MB  == first: {same}", flush=True)
