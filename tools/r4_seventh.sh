#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/ -m gpu -q -x --timeout 600 > gpurun_out/r4_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r4_gpu_tests.log
bash tools/k7_ab.sh r4_k7_pres_default variants/k7_default.so polyfuzz_amd/libpolyfuzz_hip.so | sed 's/scored/\n    scored/'
timeout 900 python tools/predict_scaling.py > gpurun_out/r4_predicted_scaling.json 2> gpurun_out/r4_predicted_scaling.err; echo "predict rc=$?"; tail -3 gpurun_out/r4_predicted_scaling.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_predicted_scaling.json'))
for c in d['configs']:
    print(c['config'][:90])
    for w,v in c.get('worlds',{}).items(): print('   N=',w, 'max/mean', v['max_over_mean'], 'job ms', v['job_ms_predicted'], 'eff', v['efficiency_predicted'])
    if 'per_rank_ms_of_2_of_8_shards' in c: print('   ', c['per_rank_ms_of_2_of_8_shards'], c['efficiency_predicted'])
PY
