#!/bin/bash
# tuning sweep of the K3 knobs (to-rows per block x waves per workgroup); prints K3 ms per step
for V in "2048 1" "2048 2" "2048 4" "4096 1" "4096 2" "4096 4" "4096 8" "8192 2" "8192 4" "8192 8"; do
  set -- $V
  echo -n "block=$1 waves=$2  "
  PFZ_K3_BLOCK=$1 PFZ_K3_WAVES=$2 python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k3_ms', d['kernel_ms_per_step']['k3_cossim_topn'], 'step_ms', round(d['ms_per_step'],3))"
done
