#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vectorize_gpu.py tests/test_random_parity_gpu.py tests/test_matchers_gpu.py -m gpu -q -x --timeout 300 > gpurun_out/r4_extract_tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r4_extract_tests.log
for w in thread wave; do
echo "== $w: $(PFZ_K1_EXTRACT=$w timeout 200 python bench.py --no-cpu-baseline --no-match-wall --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']
print('headline', round(d['ms_per_step'],3), d['kernel_ms_per_step']['k1_extract'], '| c2', round(c['c2_tfidf_10k']['ms_per_step'],3), c['c2_tfidf_10k']['kernel_ms_per_step']['k1_extract'], '| 1m', round(c['tfidf_1m_shard']['ms_per_step'],2), c['tfidf_1m_shard']['kernel_ms_per_step']['k1_extract'])")"
done
