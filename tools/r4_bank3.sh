#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_k3_cossim_gpu.py tests/test_matchers_gpu.py -m gpu -q -x --timeout 300 > gpurun_out/r4_bank_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r4_bank_tests.log
for v in 1 0; do
  if [ $v = 1 ]; then export PFZ_K3_NO_BANK_ORDER=1; else unset PFZ_K3_NO_BANK_ORDER; fi
  echo "== no_bank_order=$v: $(timeout 300 python bench.py --no-cpu-baseline --no-match-wall --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; c=d['configs']
print('headline step', round(d['ms_per_step'],3), 'k3', k['k3_cossim_topn'], 'bank', k.get('k_index_bank_order'), '| c2', round(c['c2_tfidf_10k']['ms_per_step'],3), c['c2_tfidf_10k']['kernel_ms_per_step']['k3_cossim_topn'], '| 1m', round(c['tfidf_1m_shard']['ms_per_step'],2), c['tfidf_1m_shard']['kernel_ms_per_step']['k3_cossim_topn'], c['tfidf_1m_shard']['kernel_ms_per_step'].get('k_index_bank_order'))")"
done
