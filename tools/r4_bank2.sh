#!/bin/bash
for m in 16 32 48 64 96 128; do
  echo "== PFZ_K3_BANK_MIN=$m: $(PFZ_K3_BANK_MIN=$m timeout 200 python bench.py --no-cpu-baseline --no-match-wall --no-configs --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('step', round(d['ms_per_step'],3), 'k3', k['k3_cossim_topn'], 'bank', k['k_index_bank_order'])")"
done
