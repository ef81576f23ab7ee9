#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/k3_ls_time.py 1000000 125000 4096:0:1 2048:1:4 2048:1:8 4096:1:4 2048:1:4:12 2048:1:4:16 2048:1:2:18 2>&1 | tee gpurun_out/r4_k3_ls_time2.log | grep -v "^$"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PFZ_K3_BLOCK=2048 PFZ_K3_LS_BLOCKS=4 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/ls_fetch -o ls -- python tools/k3_ls_time.py 1000000 125000 2048:1:4 > gpurun_out/ls_fetch.log 2>&1
python - <<'PY'
import sqlite3
db=sqlite3.connect('gpurun_out/ls_fetch/ls_results.db')
for r in db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '%k3_%' group by kernel_name"): print(r[0][:60], r[1], f"{r[2]*2*1024/1e9:.1f} GB (x2)")
PY
