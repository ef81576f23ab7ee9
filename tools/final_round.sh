#!/bin/bash
# End-of-round evidence in one gpurun call: GPU tests + smoke, the default bench line, the rocprofv3 summary and PMC
# passes of the same command, the EditDistance bench, the dense and the 1M-row TF-IDF shards.
# usage (on the GPU box, from the repo root): bash tools/final_round.sh <tag>
tag=${1:-final}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
tail -c 600 gpurun_out/${tag}_bench.err
timeout 300 python bench.py --config editdistance > gpurun_out/${tag}_editdistance.json 2>> gpurun_out/${tag}_bench.err
timeout 400 python tools/scale_dense_500k.py 2>&1 | tail -1 > gpurun_out/${tag}_scale_dense_500k_shard.json
timeout 400 python bench.py --config dense --steps 5 --warmup 1 > gpurun_out/${tag}_dense.json 2>> gpurun_out/${tag}_bench.err
timeout 400 python tools/scale_1m.py 2>&1 | tail -1 > gpurun_out/${tag}_scale_1m_shard.json
timeout 300 python tools/k7_time.py 20000 WRatio,partial_ratio,token_ratio > gpurun_out/${tag}_k7.log 2>&1
timeout 700 bash tools/profile_bench.sh gpurun_out/${tag}_profile > gpurun_out/${tag}_profile.log 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
for k in ("value","ms_per_step","match_wall_ms","latency","kernel_ms_per_step","parity_check"):
    print(k, d.get(k))
print({k:d["roofline"].get(k) for k in ("achieved","frac","avg_launch_ms","lds_floor_ms","frac_of_lds_floor")})
e=json.load(open("gpurun_out/${tag}_editdistance.json"))
print("edit", e["ms_per_step"], e["kernel_ms_per_step"], e["roofline"]["frac"], e["match_wall_ms"], e["match_wall_ms_to_list_resident"])
PY
cat gpurun_out/${tag}_scale_dense_500k_shard.json | cut -c1-400
cat gpurun_out/${tag}_scale_1m_shard.json | cut -c1-400
cat gpurun_out/${tag}_k7.log
ls gpurun_out/${tag}_profile | head
