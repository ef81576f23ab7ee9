#!/bin/bash
# round 5, K3 symmetric: tests of the symmetric form, headline timing, per-pass kernel times.  usage (GPU box): bash tools/r5_k3.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; T=${1:-a}; O=gpurun_out/r5_k3_$T; mkdir -p $O
timeout 900 python -m pytest tests/test_k3_cossim_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --timeout 600 -k "not edit_distance" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -5 $O/tests.log
B="python bench.py --no-configs --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall"
for i in 1 2; do timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  sym: step', round(d['ms_per_step'],4), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'], 'parity', d.get('parity_check',{}).get('ok'))"; done
PFZ_K3_SYM=0 timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  row-major: step', round(d['ms_per_step'],4), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'])"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- $B > $O/stats.log 2>&1; echo "stats rc=$?"
python tools/rocprof_summary.py $O/stats/bench_results.db 2>&1 | head -16 | cut -c1-130 | tee $O/stats_summary.txt
