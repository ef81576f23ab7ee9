#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -m gpu -q -x --timeout 600 > gpurun_out/r4_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r4_gpu_tests.log
for nto in 200000 300000 500000; do echo "n_to $nto"; timeout 300 python tools/k3_ls_time.py $nto 125000 2048:0:1 2048:1:8 2048:1:4 2>&1 | grep block; done
timeout 300 python bench.py --config tfidf_1m 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac_lds_floor'], d['parity_check'])"
