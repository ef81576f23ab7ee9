#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_k3_cossim_gpu.py -m gpu -q -x --timeout 300 -k "company or readme or knobs or lockstep" > gpurun_out/r4_bank_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r4_bank_tests.log
for rep in 1 2; do for v in 1 0; do
  if [ $v = 1 ]; then export PFZ_K3_NO_BANK_ORDER=1; else unset PFZ_K3_NO_BANK_ORDER; fi
  echo "== no_bank_order=$v: $(timeout 200 python bench.py --no-cpu-baseline --no-match-wall --no-configs --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', round(d['ms_per_step'],3), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'], 'fill', d['kernel_ms_per_step']['k_index_fill'])")"
done; done
for v in 1 0; do
  if [ $v = 1 ]; then export PFZ_K3_NO_BANK_ORDER=1; else unset PFZ_K3_NO_BANK_ORDER; fi
  echo "== 1M shard, no_bank_order=$v: $(timeout 300 python tools/k3_ls_time.py 1000000 125000 2048:1:8 2>&1 | grep block)"
done
