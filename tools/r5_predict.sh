#!/bin/bash
# round 5: the predicted 1 -> 8 curve (one GPU), with the variant library for the symmetric strong-scaling job.  usage (GPU box): bash tools/r5_predict.sh [--quick]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
POLYFUZZ_HIP_LIB=$PWD/variants/exp.so timeout 1500 python tools/predict_scaling.py $1 > gpurun_out/r05_predicted_scaling.json 2> gpurun_out/r05_predict.err
echo "rc=$?"; tail -3 gpurun_out/r05_predict.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_predicted_scaling.json'))
for c in d['configs']:
    print(c['config'][:100], c.get('single_gpu_ms'))
    for w,r in c.get('worlds',{}).items():
        print('   N=%s job %.3f ms eff %.3f max/mean %.3f'%(w, r['job_ms_predicted'], r['efficiency_predicted'], r['max_over_mean']))
PY
