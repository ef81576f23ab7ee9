#!/bin/bash
# third GPU run of K3's symmetric form: tests of the library as built, per-pass times, build variants (variants/lib_*.so).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_sym3; mkdir -p $O
timeout 400 python -m pytest tests/test_k3_cossim_gpu.py -x -q -k "symmetric" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
B="python bench.py --no-configs --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall"
one() { timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $1: step', round(d['ms_per_step'],4), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'], 'floor frac', round(d['roofline']['frac_lds_floor'],3))"; }
one default
for v in tq4 f192 push256; do POLYFUZZ_HIP_LIB=$PWD/variants/lib_$v.so one $v; done
PFZ_K3_SYM=0 one rowmajor
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- $B > $O/stats.log 2>&1; echo "stats rc=$?"
python tools/rocprof_summary.py $O/stats/bench_results.db 2>&1 | head -8 | cut -c1-130
POLYFUZZ_HIP_LIB=$PWD/variants/lib_tq4.so timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats_tq4 -o bench -- $B > $O/stats_tq4.log 2>&1
python tools/rocprof_summary.py $O/stats_tq4/bench_results.db 2>&1 | head -8 | cut -c1-130
