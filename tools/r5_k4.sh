#!/bin/bash
# round 5: K4 (EditDistance, 20k x 20k titles) -- the product and variants/*.so side by side on one box + its GPU tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for lib in "" variants/*.so; do
  [ -f "$lib" ] || [ -z "$lib" ] || continue
  if [ -n "$lib" ]; then export POLYFUZZ_HIP_LIB=$PWD/$lib; else unset POLYFUZZ_HIP_LIB; fi
  python bench.py --config editdistance --steps 20 --warmup 3 --no-cpu-baseline --no-match-wall 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[${lib:-product}] ms_per_step %.4f' % r['ms_per_step'], r.get('parity_check',{}).get('ok'))"
done; done
unset POLYFUZZ_HIP_LIB
