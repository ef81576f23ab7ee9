#!/bin/bash
# K3 what-if timing experiments (results are wrong on purpose): build one library per experiment, then
# (on the GPU box) time the headline workload with each.  usage: tools/k3_exp.sh build | run
if [ "$1" == build ]; then
  for e in 1 2 3 4; do bash tools/build_variant.sh polyfuzz_amd/libpolyfuzz_hip_exp$e.so -DPFZ_K3_EXP=$e & done; wait
else
  echo -n "product            "; python bench.py --no-cpu-baseline --no-match-wall --steps 10 --warmup 2 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms_per_step']['k3_cossim_topn'])"
  for e in "1 no-LDS-atomics" "2 no-posting-loads" "3 conflict-free-LDS" "4 no-owner-search"; do
    set -- $e
    echo -n "exp $1 $2  "; POLYFUZZ_HIP_LIB=polyfuzz_amd/libpolyfuzz_hip_exp$1.so python bench.py --no-cpu-baseline --no-match-wall --steps 10 --warmup 2 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms_per_step']['k3_cossim_topn'])"
  done
fi
