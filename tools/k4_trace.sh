#!/bin/bash
# per-launch durations of K4 on config 3 (20k x 20k IMDB titles)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/k4_trace
timeout 150 rocprofv3 --kernel-trace -d $R/gpurun_out/k4_trace -o k4 --output-format csv -- python $R/bench.py --config editdistance --steps 3 --warmup 1 --no-cpu-baseline --no-match-wall > $R/gpurun_out/k4_trace.log 2>&1
cd $R && python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/k4_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k4_' in r['Kernel_Name'] or 'best_to' in r['Kernel_Name']]
for r in rows[-8:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{r['Kernel_Name'][:70]:70s} dur {(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size', r.get('Grid_Size_X'))}")
PY
