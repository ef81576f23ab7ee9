#!/bin/bash
for v in "thread 0" "wave 16" "wave 32" "wave 64" "wave 128" "auto 0"; do set -- $v
  if [ $1 = auto ]; then unset PFZ_K1_EXTRACT PFZ_K1_WAVE_STRINGS; else export PFZ_K1_EXTRACT=$1; [ $2 != 0 ] && export PFZ_K1_WAVE_STRINGS=$2 || unset PFZ_K1_WAVE_STRINGS; fi
  echo "== $1 $2: $(timeout 100 python bench.py --config c2 --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 step', round(d['ms_per_step'],3), 'k1', d['kernel_ms_per_step']['k1_extract'], 'match', round(d.get('match_wall_ms',0),3))")"
done
