#!/bin/bash
# A/B timing of library variants on ONE box (boxes differ by ~10 % in clock): tools/k7_ab.sh <tag> a.so b.so ...
# every variant twice, alternating; WRatio and token_ratio on config 3's lists
tag=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
  for so in "$@"; do
    echo "== $so (pass $rep)" >> gpurun_out/${tag}_ab.log
    POLYFUZZ_HIP_LIB=$so timeout 200 python tools/k7_time.py 20000 WRatio,token_ratio 2>&1 | grep "20000 x" | sed 's/k7_prepare.*scored/scored/' >> gpurun_out/${tag}_ab.log
  done
done
cat gpurun_out/${tag}_ab.log
