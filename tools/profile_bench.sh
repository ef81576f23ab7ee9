#!/bin/bash
# Round profile of bench.py (run on the GPU box): kernel-trace stats + separate PMC passes for HBM traffic.
# Every pass has its own timeout: a TA-counter pass once hung for 20 minutes.
OUT=${1:-gpurun_out/profile}
# round 6: the line's default timed step is the user-level .match() call and the device-resident step (whose dominant kernel the
# roofline object times) is a second region of the same run; the passes below profile the device-resident step ALONE (--step
# device): one K3 launch per step, what `roofline.avg_launch_ms` must agree with.  stats_match: the default line, for the record.
STEP="--step device"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
# (the default line: headline + every config sub-record, CPU arms and .match() legs off -- they launch no kernel of interest)
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench -- python bench.py $STEP --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall > "$OUT/stats.log" 2>&1
# (the headline alone: the stats of the pass above mix its K3 launches with those of the 10k x 10k sub-record)
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats_headline" -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall --no-configs $STEP > "$OUT/stats_headline.log" 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-match-wall --no-configs $STEP > "$OUT/fetch.log" 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-match-wall --no-configs $STEP > "$OUT/write.log" 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace -d "$OUT/sq1" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-match-wall --no-configs $STEP > "$OUT/sq1.log" 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS --kernel-trace -d "$OUT/sq2" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-match-wall --no-configs $STEP > "$OUT/sq2.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats_match" -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall --no-configs > "$OUT/stats_match.log" 2>&1
python tools/rocprof_summary.py "$OUT"/stats_match/bench_results.db > "$OUT/summary_match.txt" 2>&1
grep '^{' "$OUT/stats.log" | tail -1 > "$OUT/bench_under_rocprof.json"
python tools/rocprof_summary.py "$OUT"/stats_headline/bench_results.db > "$OUT/summary_headline.txt" 2>&1
python tools/rocprof_summary.py "$OUT"/stats/bench_results.db "$OUT"/fetch/bench_results.db "$OUT"/write/bench_results.db "$OUT"/sq1/bench_results.db "$OUT"/sq2/bench_results.db > "$OUT/summary.txt" 2>&1
# FETCH_SIZE calibration for K3's 8-byte-per-lane posting loads (1 GiB read once per launch)
if [ -x tools/ubench/fetch_calib.bin ]; then
  timeout 100 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/calib" -o calib -- tools/ubench/fetch_calib.bin > "$OUT/calib.log" 2>&1
  python - "$OUT"/calib/calib_results.db >> "$OUT/summary.txt" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
print("== FETCH_SIZE calibration (tools/ubench/fetch_calib.hip: 1 GiB = 1048576 KB read once per launch)")
for r in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    print(f"  {r[0][:40]:40s} {r[1]:12s} {r[2]:4d} launches  avg {r[3]:14.1f}  -> reported / actual = {r[3] / 1048576.0:.3f}")
PY
fi
