#!/bin/bash
# round 5: the kernel timeline of one config-2 step (10k x 10k) -- where the device idles between launches.  usage (GPU box): bash tools/r5_c2_trace.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_c2_trace; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/t -o c2 -- python bench.py --config c2 --steps 6 --warmup 2 --no-cpu-baseline --no-match-wall > $O/log.txt 2>&1; echo rc=$?
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/r5_c2_trace/t/*.db')[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t]
print(kd[:5])
t = 'kernels' if 'kernels' in tabs else kd[0]
cols = [r[1] for r in db.execute(f'pragma table_info({t})')]
print(cols)
PY
