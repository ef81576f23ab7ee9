import json, sys, os, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polyfuzz_amd import _lib
import polyfuzz_amd
ctx = polyfuzz_amd.Context(0)
g = json.load(open("tests/golden/company_c2_lists.json"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
fl, tl = g["from_list"][:n], g["to_list"][:n]
print("maxlen", max(map(len, fl)), flush=True)
f = _lib.DeviceStrings.upload(ctx, fl); print("up f", flush=True)
t = _lib.DeviceStrings.upload(ctx, tl); print("up t", flush=True)
vec = _lib.DeviceTfidf.fit(ctx, _lib.TfidfParams(3, 3, 1, 1), t, f); ctx.sync(); print("fit", vec.info(), flush=True)
a = vec.transform(f); ctx.sync(); print("tr f", a.shape, flush=True)
b = vec.transform(t); ctx.sync(); print("tr t", b.shape, flush=True)
x = a.download(); print("dl", x[0][-1], flush=True)
