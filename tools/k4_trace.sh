#!/bin/bash
# per-kernel durations of one EditDistance bench run (config 3)
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace -d $R/gpurun_out/k4_trace -o k4 --output-format csv -- python $R/bench.py --config editdistance --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/k4_trace.log 2>&1
cd $R && python - <<'PY'
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/k4_trace/k4_kernel_trace.csv')):
    d[r['Kernel_Name'][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    print('%-72s n=%3d avg %8.1f us  min %8.1f' % (k, len(v), sum(v) / len(v), min(v)))
PY
