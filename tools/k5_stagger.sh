export K5_FILLS=1
for ns in 0 5000 10000 15000 21300 26000 31000; do echo "== stagger $ns ns"; PFZ_K5_STAGGER_NS=$ns python tools/k5_gemm_time.py; done
echo "== default"; python tools/k5_gemm_time.py
python -m pytest tests/test_dense_gpu.py -q -x 2>&1 | tail -2
