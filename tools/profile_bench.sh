#!/bin/bash
# Round profile of bench.py (run on the GPU box): kernel-trace stats + separate PMC passes for HBM traffic.
# Every pass has its own timeout: a TA-counter pass once hung for 20 minutes.
OUT=${1:-gpurun_out/profile}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/stats.log" 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/fetch.log" 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/write.log" 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace -d "$OUT/sq1" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/sq1.log" 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS --kernel-trace -d "$OUT/sq2" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/sq2.log" 2>&1
grep '^{' "$OUT/stats.log" | tail -1 > "$OUT/bench_under_rocprof.json"
python tools/rocprof_summary.py "$OUT"/stats/bench_results.db "$OUT"/fetch/bench_results.db "$OUT"/write/bench_results.db "$OUT"/sq1/bench_results.db "$OUT"/sq2/bench_results.db > "$OUT/summary.txt" 2>&1
