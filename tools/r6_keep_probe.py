"""round 6: TFIDF.match(names) timed as bench.py times it -- K calls back to back, their frames KEPT on a warmed heap -- against
the per-call clock of tools/r6_match_ab.py (the previous frame dropped before the clock starts), for the host-thread settings given
as NAME:ENV=v,ENV=v (as r6_match_ab.py).  Says where the two clocks differ: stage stamps of the median call of either kind."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from polyfuzz_amd import datasets, _lib
from polyfuzz_amd.models import TFIDF, _utils, _tfidf
names = datasets.load_company_names()
K = 20
bench.FrameKeeper.warm(len(names), 5, 3 * K + 8)
variants = []
for a in sys.argv[1:] or ["default:"]:
    name, _, envs = a.partition(":")
    variants.append((name, {kv.split("=")[0]: kv.split("=")[1] for kv in envs.split(",") if kv}))
KEYS = ("PFZ_RANGE_FILL", "PFZ_RANGE_THREADS", "PFZ_PACK_INTO_THREADS", "PFZ_HOST_THREADS", "PFZ_HOST_PIN")
m = TFIDF(min_similarity=0, top_n=5)
for name, env in variants:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    _tfidf._RANGE_FILL = os.environ.get("PFZ_RANGE_FILL", "1") != "0"
    # (PFZ_RANGE_THREADS / PFZ_PACK_INTO_THREADS are names of THIS tool: the two crews' sizes apart; the library reads PFZ_HOST_THREADS for both)
    _utils._RANGE_THREADS = int(os.environ.get("PFZ_RANGE_THREADS", _lib.host_threads()))
    _lib._PACK_INTO_THREADS = int(os.environ.get("PFZ_PACK_INTO_THREADS", _lib.host_threads()))
    for _ in range(4):
        m.match(names)
    for mode in ("kept", "dropped"):
        keep, ts, st = [], [], []
        df = None
        for _ in range(K):
            if mode == "dropped":
                df = None
            t0 = time.perf_counter()
            df = m.match(names)
            ts.append((time.perf_counter() - t0) * 1e3)
            st.append(m.last_timings)
            if mode == "kept":
                keep.append(df)
        i = int(np.argsort(ts)[len(ts) // 2])
        print(f"{name:12s} {mode:8s}: median {ts[i]:.3f} ms (min {min(ts):.3f}, mean {np.mean(ts):.3f}); stages " +
              ", ".join(f"{k} {v:.2f}" for k, v in st[i].items()), flush=True)
        del keep
