#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_k3_cossim_gpu.py -m gpu -q -x --timeout 300 > gpurun_out/r4_third_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r4_third_tests.log
timeout 600 python tools/k3_ls_time.py 2>&1 | tee gpurun_out/r4_k3_ls_time.log | grep -v "^$"
