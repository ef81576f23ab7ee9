#!/bin/bash
mkdir -p gpurun_out
for v in "0 1" "1 2" "1 4" "1 8"; do set -- $v
  echo "== headline, PFZ_K3_LOCKSTEP=$1 PFZ_K3_LS_BLOCKS=$2"
  PFZ_K3_LOCKSTEP=$1 PFZ_K3_LS_BLOCKS=$2 timeout 200 python bench.py --no-cpu-baseline --no-match-wall --no-configs --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step']['k3_cossim_topn'])"
done
echo "== c2"; for v in "0 1" "1 2"; do set -- $v
  PFZ_K3_LOCKSTEP=$1 PFZ_K3_LS_BLOCKS=$2 timeout 200 python bench.py --config c2 --no-cpu-baseline --no-match-wall --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step']['k3_cossim_topn'])"
done
