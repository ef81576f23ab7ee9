#!/bin/bash
# fourth GPU run: DPP wave maximum (all K3 kernels), merge by size, pass-0 extension of weak rows
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_sym4; mkdir -p $O
timeout 600 python -m pytest tests/test_k3_cossim_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
B="python bench.py --no-configs --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall"
one() { timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $1: step', round(d['ms_per_step'],4), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'], 'floor frac', round(d['roofline']['frac_lds_floor'],3))"; }
one default
PFZ_K3_SYM_NO_EXT=1 one no_ext
POLYFUZZ_HIP_LIB=$PWD/variants/lib_shfl.so one shfl_max
PFZ_K3_SYM=0 one rowmajor_dpp
PFZ_K3_SYM=0 POLYFUZZ_HIP_LIB=$PWD/variants/lib_shfl.so one rowmajor_shfl
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- $B > $O/stats.log 2>&1; echo "stats rc=$?"
python tools/rocprof_summary.py $O/stats/bench_results.db 2>&1 | head -8 | cut -c1-130
timeout 100 python bench.py --config tfidf_1m --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  1m shard: step', round(d['ms_per_step'],3), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'])"
timeout 100 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  c2: step', round(d['ms_per_step'],4), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'])"
