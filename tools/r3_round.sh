#!/bin/bash
# One gpurun call of round 3's build -> measure loop: GPU tests, then the default bench line (headline + every config).
# usage (on the GPU box, from the repo root): bash tools/r3_round.sh <tag> [extra pytest args]
tag=${1:-r3}; shift
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x --timeout 240 "$@" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
tail -15 gpurun_out/${tag}_tests.log
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
grep -v -E "^\s" gpurun_out/${tag}_bench.err | tail -c 1500
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
for k in ("value","ms_per_step","match_wall_ms","match_pairs_per_s","latency","kernel_ms_per_step","parity_check","cpu_baseline"):
    print(k, d.get(k))
print({k:d["roofline"].get(k) for k in ("achieved","peak","frac","avg_launch_ms","lds_floor_ms","hbm_priced_frac")})
for name, c in d.get("configs", {}).items():
    print("==", name, {k: c.get(k) for k in ("error","ms_per_step","match_wall_ms","match_wall_ms_to_list_resident","kernel_ms_per_step","bench_wall_s")})
    print("   roofline", {k: (c.get("roofline") or {}).get(k) for k in ("kernel","achieved","peak","frac","unit")})
    print("   cpu", (c.get("cpu_baseline") or {}).get("value"), "parity", c.get("parity_check"))
PY
