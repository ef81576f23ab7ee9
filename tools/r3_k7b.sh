#!/bin/bash
# K7 iteration on the GPU box, short form: the scorer parity tests, then timings.  usage: bash tools/r3_k7b.sh <tag> [names]
tag=${1:-k7}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x --timeout 200 > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
tail -5 gpurun_out/${tag}_tests.log
timeout 300 python tools/k7_time.py 20000 WRatio,partial_ratio,token_ratio,partial_token_ratio $2 > gpurun_out/${tag}_time.log 2>&1
tail -12 gpurun_out/${tag}_time.log
PFZ_K7_EXP=1 timeout 100 python tools/k7_time.py 20000 WRatio > gpurun_out/${tag}_sweeponly.log 2>&1
tail -2 gpurun_out/${tag}_sweeponly.log
