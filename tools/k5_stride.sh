#!/bin/bash
export K5_FILLS=1 PFZ_K5_STAGGER_NS=0
for n in 32768 33024 33280 36864; do echo "== n=$n shipped"; python tools/k5_gemm_time.py $n; echo "== n=$n 1 WG/CU"; PFZ_K5_ONE_WG=1 python tools/k5_gemm_time.py $n; 
echo "== n=$n EXP4"; POLYFUZZ_HIP_LIB=$PWD/polyfuzz_amd/_k5exp4.so python tools/k5_gemm_time.py $n; done
echo "== n=33024 stagger default"; env -u PFZ_K5_STAGGER_NS python tools/k5_gemm_time.py 33024
