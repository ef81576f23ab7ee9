#!/bin/bash
export K5_FILLS=1
echo "== block map, stage at 8"; python tools/k5_gemm_time.py
echo "== linear map, stage at 8"; PFZ_K5_LINEAR_MAP=1 python tools/k5_gemm_time.py
echo "== block map, stage at 24"; POLYFUZZ_HIP_LIB=$PWD/polyfuzz_amd/_k5st24.so python tools/k5_gemm_time.py
echo "== linear map, stage at 24"; PFZ_K5_LINEAR_MAP=1 POLYFUZZ_HIP_LIB=$PWD/polyfuzz_amd/_k5st24.so python tools/k5_gemm_time.py
python -m pytest tests/test_dense_gpu.py -q -x 2>&1 | tail -2
