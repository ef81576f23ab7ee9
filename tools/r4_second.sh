#!/bin/bash
# round 4, second GPU call: the new parity tests, the PMC traffic of K3 on the three TF-IDF workloads, one bench line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_indel_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --timeout 400 > gpurun_out/r4_second_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r4_second_tests.log
bash tools/pmc_traffic.sh gpurun_out/pmc_traffic 2>&1 | tail -70
timeout 400 python bench.py > gpurun_out/r4_second_bench.json 2> gpurun_out/r4_second_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_second_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','match_wall_ms') if k in d})
print(json.dumps(d['roofline'],indent=0)[:1500]); print(json.dumps(d['parity_check'],indent=0))
for k,v in d.get('configs',{}).items(): print(k, v.get('ms_per_step'), v.get('kernel_ms_per_step'), v.get('roofline',{}).get('frac'), v.get('roofline',{}).get('frac_lds_floor'), v.get('parity_check',{}).get('ok'))
PY
