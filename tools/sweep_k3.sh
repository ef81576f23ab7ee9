#!/bin/bash
# tuning sweep of the K3 knob (to-rows per block); prints K3 ms per step on the headline workload
for B in 1024 1536 2048 4096; do
  echo -n "block=$B  "
  PFZ_K3_BLOCK=$B python bench.py --no-cpu-baseline --no-match-wall --steps 10 --warmup 2 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k3_ms', d['kernel_ms_per_step']['k3_cossim_topn'], 'step_ms', round(d['ms_per_step'],3), 'index_fill', d['kernel_ms_per_step']['k_index_fill'])"
done
