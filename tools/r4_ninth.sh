#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/match_split_probe.py 2>&1 | tee gpurun_out/r4_match_split.log
echo "== K7 few rows with the new parts rule"
for n in 20000 10000 5000 2500 1000; do
  echo "n_from $n: $(timeout 100 python tools/k7_time.py $n WRatio 2>&1 | grep ' x ' | sed 's/k7_prepare.*//')"
done
timeout 300 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -2
