#!/bin/bash
# round 5: K7 timings (product and variants/*.so side by side on one box) + its GPU tests.  usage (GPU box): bash tools/r5_k7.sh [modes]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
M=${1:-WRatio,partial_ratio,token_ratio,partial_token_ratio}
for rep in 1 2; do for lib in "" variants/*.so; do
  [ -f "$lib" ] || [ -z "$lib" ] || continue
  if [ -n "$lib" ]; then export POLYFUZZ_HIP_LIB=$PWD/$lib; else unset POLYFUZZ_HIP_LIB; fi
  python tools/k7_time.py 20000 $M 2>&1 | grep "20000 x" | cut -c1-118 | sed "s|^|[${lib:-product}] |"
done; done
unset POLYFUZZ_HIP_LIB
timeout 900 python -m pytest tests/test_fuzz_gpu.py tests/test_bench_local_gpu.py -m gpu -q --timeout 600 -k "not tfidf" 2>&1 | tail -2
