"""round 6: TFIDF(min_similarity=0, top_n=5).match(names) on ONE box, in ONE process, alternating variants given as
NAME:ENV=VALUE,ENV=VALUE ... (PFZ_MATCH_SHARES uses ';' between shares).  Defaults: the round-5 form (a session launch per range,
PFZ_K3_NO_STREAMED=1, four ranges) against the streamed session with several range cuts.  No profiler, no kernel timers: wall clock
and the matcher's own stage stamps; the previous frame is dropped BEFORE the clock starts.
usage: python tools/r6_match_ab.py [name:ENV=v,ENV=v ...]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polyfuzz_amd import datasets, _lib
from polyfuzz_amd.models import TFIDF, _utils, _tfidf
names = datasets.load_company_names()
KEYS = ("PFZ_K3_NO_STREAMED", "PFZ_MATCH_SHARES", "PFZ_DIRECT_PACK", "PFZ_RANGE_FILL", "PFZ_RANGE_THREADS", "PFZ_PACK_INTO_THREADS", "PFZ_HOST_THREADS", "PFZ_HOST_PIN")
variants = [("r5form", {"PFZ_K3_NO_STREAMED": "1", "PFZ_MATCH_SHARES": "0.3,0.3,0.25,0.15"}),
            ("streamed5", {"PFZ_MATCH_SHARES": "0.2,0.2,0.2,0.2,0.2"}),
            ("streamed4", {"PFZ_MATCH_SHARES": "0.25,0.3,0.25,0.2"}),
            ("streamed8", {"PFZ_MATCH_SHARES": "0.1,0.15,0.15,0.15,0.15,0.1,0.1,0.1"}),
            ("streamed10", {"PFZ_MATCH_SHARES": "0.1,0.1,0.1,0.1,0.1,0.1,0.1,0.1,0.1,0.1"}),
            ("streamed6", {"PFZ_MATCH_SHARES": "0.15,0.2,0.2,0.2,0.15,0.1"}),
            ("streamed default", {})]
if len(sys.argv) > 1:
    variants = []
    for a in sys.argv[1:]:
        name, _, envs = a.partition(":")
        variants.append((name, {kv.split("=")[0]: kv.split("=")[1].replace(";", ",") for kv in envs.split(",") if kv}))
def apply_env():
    # (knobs read at import: set as the variant says)
    _tfidf._DIRECT_PACK = os.environ.get("PFZ_DIRECT_PACK", "1") != "0"
    _tfidf._RANGE_FILL = os.environ.get("PFZ_RANGE_FILL", "1") != "0"
    # (PFZ_RANGE_THREADS / PFZ_PACK_INTO_THREADS are names of THIS tool: the two crews' sizes apart; the library reads PFZ_HOST_THREADS for both)
    _utils._RANGE_THREADS = int(os.environ.get("PFZ_RANGE_THREADS", _lib.host_threads()))
    _lib._PACK_INTO_THREADS = int(os.environ.get("PFZ_PACK_INTO_THREADS", _lib.host_threads()))


first_sig = None
m = TFIDF(min_similarity=0, top_n=5)
res = {k: ([], []) for k, _ in variants}
df = None
for rep in range(16):
    for name, env in variants:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        apply_env()
        df = None
        t0 = time.perf_counter(); df = m.match(names); dt = (time.perf_counter() - t0) * 1e3
        if rep == 0:         # (every variant returns the same frame)
            sig = (df["To"].tolist(), df["To_5"].tolist(), df["Similarity_3"].to_numpy().copy())
            if first_sig is None:
                first_sig = sig
            assert sig[0] == first_sig[0] and sig[1] == first_sig[1] and np.array_equal(sig[2], first_sig[2]), name
            del sig
        if rep >= 3:
            res[name][0].append(dt); res[name][1].append(m.last_timings)
if os.environ.get("PFZ_MATCH_TRACE"):
    for name, env in variants:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        apply_env()
        df = None
        t0 = time.perf_counter(); df = m.match(names); dt = (time.perf_counter() - t0) * 1e3
        print(name, f"{dt:.3f} ms:", "; ".join(f"{k} {v * 1e3:.2f}" for k, v in (m.last_trace or [])))
for name, _ in variants:
    ts, st = res[name]
    i = int(np.argsort(ts)[len(ts) // 2])
    print(f"{name:44s}: wall median {ts[i]:.3f} ms (min {min(ts):.3f}); stages " + ", ".join(f"{k} {v:.2f}" for k, v in st[i].items()), flush=True)
