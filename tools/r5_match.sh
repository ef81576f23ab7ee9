#!/bin/bash
# round 5: .match() wall of the headline under environment settings, interleaved (3 rounds) on one box.
# usage (GPU box): bash tools/r5_match.sh "PFZ_X=0" "PFZ_Y=1" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for round in 1 2 3; do for e in "$@"; do echo -n "[$e] "; env $e timeout 300 python - <<'PY'
import os, time, json, gc
from polyfuzz_amd import datasets
from polyfuzz_amd.models import TFIDF
names = datasets.load_company_names()
m = TFIDF(min_similarity=0, top_n=5)
for _ in range(4): df = m.match(names)
ts, st = [], []
for _ in range(21):
    df = None
    t0 = time.perf_counter(); df = m.match(names); ts.append((time.perf_counter() - t0) * 1e3); st.append(m.last_timings)
o = sorted(range(len(ts)), key=lambda i: ts[i]); med = o[len(o) // 2]
print('median %.3f min %.3f' % (ts[med], min(ts)), {k: round(v, 3) for k, v in st[med].items()})
PY
done; done
