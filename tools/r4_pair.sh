#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_k3_cossim_gpu.py -m gpu -q -x --timeout 300 -k "pair_kernel or company or readme" > gpurun_out/r4_pair_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r4_pair_tests.log
for v in 0 1; do
  echo "== headline PFZ_K3_PAIR=$v: $(PFZ_K3_PAIR=$v timeout 200 python bench.py --no-match-wall --no-configs --steps 10 --cpu-seconds 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step']['k3_cossim_topn'], d['parity_check']['ok'], d['parity_check']['rows_with_index_diff'], d['parity_check']['max_abs_score_err'])")"
done
