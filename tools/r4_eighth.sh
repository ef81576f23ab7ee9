#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_local_gpu.py tests/test_comm_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --timeout 600 > gpurun_out/r4_eighth_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r4_eighth_tests.log
echo "== K7 presence by mode (new default) vs round-3 default"
bash tools/k7_ab.sh r4_k7_mode variants/k7_default.so polyfuzz_amd/libpolyfuzz_hip.so | grep -v "^$" | sed 's/scored.*//'
echo "== K7 with few from-rows: default parts vs PFZ_K7_PARTS=1"
for n in 10000 5000 2500 1000; do
  echo "n_from $n default: $(timeout 100 python tools/k7_time.py $n WRatio 2>&1 | grep ' x ' | sed 's/k7_prepare.*//')"
  echo "n_from $n parts=1: $(PFZ_K7_PARTS=1 timeout 100 python tools/k7_time.py $n WRatio 2>&1 | grep ' x ' | sed 's/k7_prepare.*//')"
done
echo "== headline with the lock-step kernel forced"
for v in "0 1" "1 4" "1 8"; do set -- $v
  echo "LOCKSTEP=$1 S=$2: $(PFZ_K3_LOCKSTEP=$1 PFZ_K3_LS_BLOCKS=$2 timeout 200 python bench.py --no-cpu-baseline --no-match-wall --no-configs --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step']['k3_cossim_topn'])")"
done
echo "== match wall"
timeout 300 python bench.py --no-cpu-baseline --no-configs --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['match_wall_ms'], d.get('match_stages_ms'), d['latency'])"
