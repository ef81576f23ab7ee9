timeout 300 python -m pytest tests/test_k3_cossim_gpu.py tests/test_fullsize_gpu.py tests/test_random_parity_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
for i in 1 2; do python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms_per_step']['k3_cossim_topn'], d['ms_per_step'])"; done
