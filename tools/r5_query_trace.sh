#!/bin/bash
# round 5: the device timeline of single-query matches against the fitted 100k list.  usage (GPU box): bash tools/r5_query_trace.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_query_trace; rm -rf $O; mkdir -p $O
cat > $O/q.py <<'PY'
import time
from polyfuzz_amd import datasets
from polyfuzz_amd.models import TFIDF
names = datasets.load_company_names()
m = TFIDF(min_similarity=0, top_n=1)
m.match(names[:1000], names)
q = [names[50000]]
for _ in range(20): m.match(q, names, re_train=False)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); m.match(q, names, re_train=False); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort(); print('single query median %.4f ms min %.4f' % (ts[100], ts[0]))
PY
PYTHONPATH=$GRAFT_REPO_ROOT python $O/q.py
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/t -o q -- env PYTHONPATH=$GRAFT_REPO_ROOT python $O/q.py > $O/log.txt 2>&1; echo rc=$?
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/r5_query_trace/t/*.db')[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
ev = [(s,e,n[:60]) for n,s,e in rows]
idx=[i for i,x in enumerate(ev) if 'k_extract' in x[2]]
seg=ev[idx[-2]:idx[-1]]
t0=seg[0][0]; pe=None
for s,e,n in seg:
    print('%8.1f %7.1f us  gap %6.1f  %s'%((s-t0)/1e3,(e-s)/1e3,(s-pe)/1e3 if pe else 0,n)); pe=e
PY
