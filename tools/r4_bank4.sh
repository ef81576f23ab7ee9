#!/bin/bash
for v in 1 0; do
  if [ $v = 1 ]; then export PFZ_K3_NO_BANK_ORDER=1; else unset PFZ_K3_NO_BANK_ORDER; fi
  echo "== c2 no_bank_order=$v: $(timeout 100 python bench.py --config c2 --no-cpu-baseline --no-match-wall --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['kernel_ms_per_step'])")"
done
