#!/bin/bash
# tuning sweep of K3's pruning knobs (heavy slots per to-row x alpha in percent); prints K3 ms per step
for V in "0 0" "32 50" "32 60" "32 75" "32 90" "64 60" "64 75" "64 90" "16 75"; do
  set -- $V
  echo -n "heavy=$1 alpha=$2  "
  PFZ_K3_HEAVY=$1 PFZ_K3_ALPHA=$2 python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('k3_ms', k['k3_cossim_topn'], 'fill', k.get('k_index_fill'), 'sel', k.get('k_heavy_select'), 'step_ms', round(d['ms_per_step'],3))"
done
