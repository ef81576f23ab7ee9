#!/bin/bash
# rocprofv3 --kernel-trace --stats of the other kernels' workloads (run on the GPU box): K4 on config 3, K5 on the
# 62.5k x 500k x 768 shard, K7 (WRatio) on config 3's lists.  Summaries -> <out>/summary_other.txt
OUT=${1:-gpurun_out/profile_other}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/k4" -o run -- python bench.py --config editdistance --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall > "$OUT/k4.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/k5" -o run -- python tools/scale_dense_500k.py > "$OUT/k5.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/k7" -o run -- python tools/k7_time.py 20000 WRatio > "$OUT/k7.log" 2>&1
python tools/rocprof_summary.py "$OUT"/k4/run_results.db "$OUT"/k5/run_results.db "$OUT"/k7/run_results.db > "$OUT/summary_other.txt" 2>&1
tail -2 "$OUT/k4.log" | cut -c1-300; tail -1 "$OUT/k5.log" | cut -c1-300; tail -1 "$OUT/k7.log"
cat "$OUT/summary_other.txt" | cut -c1-150 | head -60
