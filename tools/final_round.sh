#!/bin/bash
# End-of-round evidence in one gpurun call: GPU tests + smoke, the default bench line (headline + every config sub-record),
# the rocprofv3 summary and PMC passes of the same command, K7's timings and per-launch trace.
# usage (on the GPU box, from the repo root): bash tools/final_round.sh <tag>
tag=${1:-final}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
grep -E "passed|failed|error" gpurun_out/${tag}_tests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
tail -c 600 gpurun_out/${tag}_bench.err
timeout 300 python tools/k7_time.py 20000 WRatio,partial_ratio,token_ratio,partial_token_ratio names > gpurun_out/${tag}_k7.log 2>&1
timeout 200 bash tools/k7_trace.sh >> gpurun_out/${tag}_k7.log 2>&1
timeout 200 python tools/k7_rowstats.py WRatio 2>&1 | grep -E "as shipped|wave time|timeline|scoring batches|began" >> gpurun_out/${tag}_k7.log
timeout 900 bash tools/profile_bench.sh gpurun_out/${tag}_profile > gpurun_out/${tag}_profile.log 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
for k in ("value","ms_per_step","match_wall_ms","match_pairs_per_s","latency","kernel_ms_per_step","parity_check"):
    print(k, d.get(k))
print({k:d["roofline"].get(k) for k in ("achieved","peak","frac","avg_launch_ms","lds_floor_ms","hbm_priced_frac")})
for name, c in d.get("configs", {}).items():
    print("==", name, {k: c.get(k) for k in ("error","ms_per_step","match_wall_ms","match_wall_ms_to_list_resident","kernel_ms_per_step","bench_wall_s")})
    print("   roofline", {k: (c.get("roofline") or {}).get(k) for k in ("kernel","achieved","peak","frac","unit","scored_fraction")})
    print("   cpu", (c.get("cpu_baseline") or {}).get("value"), "parity", c.get("parity_check"))
PY
cat gpurun_out/${tag}_k7.log | tail -14
head -40 gpurun_out/${tag}_profile/summary.txt
