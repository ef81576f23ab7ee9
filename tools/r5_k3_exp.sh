#!/bin/bash
# round 5: what-if builds of K3's symmetric pass 1 (variants/sym_exp*.so, tools/build_variant.sh -DPFZ_K3_SYM_EXP=n; results wrong on purpose)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --no-configs --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall"
run() { timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $1: step', round(d['ms_per_step'],4), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'])"; }
run product
for v in variants/*.so; do POLYFUZZ_HIP_LIB=$PWD/$v run $v; done
run product-again
