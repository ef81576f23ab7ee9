#!/bin/bash
# fifth GPU run: pass 2 in slices, 256 push slots again, DPP warm start
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_sym5; mkdir -p $O
timeout 600 python -m pytest tests/test_k3_cossim_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
B="python bench.py --no-configs --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall"
one() { timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $1: step', round(d['ms_per_step'],4), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'], 'floor frac', round(d['roofline']['frac_lds_floor'],3))"; }
one default
POLYFUZZ_HIP_LIB=$PWD/variants/lib_push1024.so one push1024
PFZ_K3_SYM=0 one rowmajor
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- $B > $O/stats.log 2>&1; echo "stats rc=$?"
python tools/rocprof_summary.py $O/stats/bench_results.db 2>&1 | head -9 | cut -c1-130
timeout 100 python bench.py --no-configs --steps 5 --warmup 2 --cpu-seconds 2 --no-match-wall 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  parity', d['parity_check']['ok'], d['parity_check']['rows_checked'], d['parity_check']['index_diffs_not_near_ties'])"
