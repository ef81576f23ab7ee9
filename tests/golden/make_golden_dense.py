"""Golden fixture for the dense (embedding) branch: the reference's own unit-norm 300-d fixtures
(/root/reference/tests/from_list.npy, to_list.npy; used by its tests/models/test_utils.py:9-35 and
test_embeddings.py) and the DataFrames the REFERENCE's cosine_similarity(..., method="sklearn")
produces from them.  Run in the build container only; output: tests/golden/dense_golden.npz + .json."""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
for name in ("seaborn",):
    sys.modules[name] = types.ModuleType(name)
rf = types.ModuleType("rapidfuzz"); rf.fuzz = types.ModuleType("rapidfuzz.fuzz"); rf.process = types.ModuleType("rapidfuzz.process")
rf.fuzz.ratio = rf.fuzz.WRatio = lambda a, b, **k: 0.0
sys.modules.update({"rapidfuzz": rf, "rapidfuzz.fuzz": rf.fuzz, "rapidfuzz.process": rf.process})
from polyfuzz.models._utils import cosine_similarity  # noqa: E402

fl = ["apple", "apples", "appl", "recal", "house", "similarity"]
tl = ["apple", "apples", "mouse"]
a = np.load(os.path.join(REF, "tests", "from_list.npy"))
b = np.load(os.path.join(REF, "tests", "to_list.npy"))
out = {"from_list": fl, "to_list": tl, "cases": []}
for top_n in (1, 2, 3):
    df = cosine_similarity(a, b, fl, tl, min_similarity=0.0, top_n=top_n, method="sklearn")
    out["cases"].append({"top_n": top_n, "self": False,
                         "df": {c: [None if v is None or v != v else v for v in df[c].tolist()] for c in df.columns}})
df = cosine_similarity(a, a, fl, None, min_similarity=0.0, top_n=2, method="sklearn")
out["cases"].append({"top_n": 2, "self": True,
                     "df": {c: [None if v is None or v != v else v for v in df[c].tolist()] for c in df.columns}})
np.savez_compressed(os.path.join(HERE, "dense_golden.npz"), from_vec=a, to_vec=b)
json.dump(out, open(os.path.join(HERE, "dense_golden.json"), "w"), indent=1)
print("ok", a.shape, b.shape)
