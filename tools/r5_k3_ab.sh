#!/bin/bash
# round 5: the headline's K3 -- the product and variants/*.so side by side on one box (three rounds), then the K3 tests on the product
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --no-configs --steps 12 --warmup 3 --no-cpu-baseline --no-match-wall"
for rep in 1 2 3; do for lib in "" variants/*.so; do
  [ -f "$lib" ] || [ -z "$lib" ] || continue
  if [ -n "$lib" ]; then export POLYFUZZ_HIP_LIB=$PWD/$lib; else unset POLYFUZZ_HIP_LIB; fi
  timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[${lib:-product}] step', round(d['ms_per_step'],4), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'])"
done; done
unset POLYFUZZ_HIP_LIB
[ "$1" = "notest" ] || timeout 900 python -m pytest tests/test_k3_cossim_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --timeout 600 -k "not edit_distance" 2>&1 | tail -2
