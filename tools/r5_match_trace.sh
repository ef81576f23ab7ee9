#!/bin/bash
# round 5: the device timeline of one TFIDF(top_n=5).match(names) on the headline list, next to the host's stage stamps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_match_trace; rm -rf $O; mkdir -p $O
cat > $O/m.py <<'PY'
import time
from polyfuzz_amd import datasets
from polyfuzz_amd.models import TFIDF
names = datasets.load_company_names()
m = TFIDF(min_similarity=0, top_n=5)
for _ in range(4): m.match(names)
ts = []
for _ in range(9):
    t0 = time.perf_counter(); m.match(names); ts.append(((time.perf_counter() - t0) * 1e3, m.last_timings))
ts.sort(key=lambda x: x[0]); print('match median %.3f ms' % ts[4][0], {k: round(v, 3) for k, v in ts[4][1].items()})
PY
PYTHONPATH=$GRAFT_REPO_ROOT python $O/m.py
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/t -o m -- env PYTHONPATH=$GRAFT_REPO_ROOT python $O/m.py > $O/log.txt 2>&1; echo rc=$?; grep "match median" $O/log.txt
python tools/r6_trace_parse.py; exit 0
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/r5_match_trace/t/*.db')[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
try:
    cp = db.execute("select name, start, end from memory_copies order by start").fetchall()
except Exception as e:
    cp = []
ev = sorted([(s, e, n[:64]) for n, s, e in rows] + [(s, e, 'COPY ' + str(n)[:40]) for n, s, e in cp])
idx = [i for i, x in enumerate(ev) if 'k_alpha_mark' in x[2] or ('k_extract' in x[2] and 'true' in x[2])]
first = [i for i in idx if i == 0 or (ev[i][0] - ev[i - 1][1]) > 200e3]      # a match starts after a long pause
a, b = first[-2], first[-1]
t0 = ev[a][0]; pe = None
for s, e, n in ev[a - 2:b]:
    print('%8.1f %7.1f us  gap %6.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - pe) / 1e3 if pe else 0, n)); pe = e
PY
