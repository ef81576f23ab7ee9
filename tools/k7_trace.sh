#!/bin/bash
# per-launch durations of K7 (WRatio, IMDB 20k x 20k) from rocprofv3 --kernel-trace
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/k7_trace
timeout 150 rocprofv3 --kernel-trace -d $R/gpurun_out/k7_trace -o k7 --output-format csv -- python $R/tools/k7_time.py 20000 WRatio > $R/gpurun_out/k7_trace.log 2>&1
cd $R && python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/k7_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k7_' in r['Kernel_Name']]
t0 = None
for r in rows[-14:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    t0 = t0 or s
    print(f"{r['Kernel_Name'][:60]:60s} start {(s - t0) / 1e6:8.3f} ms  dur {(e - s) / 1e6:8.3f} ms  grid {r.get('Grid_Size', r.get('Grid_Size_X'))} wg {r.get('Workgroup_Size', r.get('Workgroup_Size_X'))}")
PY
