#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_tmp; mkdir -p $O
timeout 400 python -m pytest tests/test_k3_cossim_gpu.py -x -q -k "symmetric" 2>&1 | tail -2
B="python bench.py --no-configs --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall"
one() { timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $1: step', round(d['ms_per_step'],4), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'], 'floor frac', round(d['roofline']['frac_lds_floor'],3))"; }
one default
one default_again
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- $B > $O/stats.log 2>&1
python tools/rocprof_summary.py $O/stats/bench_results.db 2>&1 | grep "k3_sym" | cut -c1-130
