#!/bin/bash
# End-of-round evidence (second half of round 4, with K3's symmetric form) in one gpurun call: smoke, the tests around the
# frame / matcher changes, the default bench line, rocprofv3 stats + PMC passes of the same command (tools/profile_bench.sh),
# the headline record of profiles/k3_hbm_traffic.json from those passes.  usage (GPU box): bash tools/r4_final.sh [tag]
tag=${1:-r04b}
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_matchers_gpu.py tests/test_fullsize_gpu.py tests/test_facade_flow_gpu.py tests/test_k3_cossim_gpu.py tests/test_fuzz_gpu.py tests/test_vectorize_gpu.py tests/test_comm_gpu.py -q 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; tail -c 400 gpurun_out/${tag}_bench.err
timeout 900 bash tools/profile_bench.sh gpurun_out/${tag}_profile > gpurun_out/${tag}_profile.log 2>&1
# the headline's traffic record from the FETCH / WRITE passes above (the c2 and 1M-shard records of the first half stay)
mkdir -p gpurun_out/${tag}_pmc && rm -rf gpurun_out/${tag}_pmc/tfidf_FETCH_SIZE gpurun_out/${tag}_pmc/tfidf_WRITE_SIZE
cp -r gpurun_out/${tag}_profile/fetch gpurun_out/${tag}_pmc/tfidf_FETCH_SIZE; cp -r gpurun_out/${tag}_profile/write gpurun_out/${tag}_pmc/tfidf_WRITE_SIZE
python tools/pmc_traffic.py gpurun_out/${tag}_pmc > gpurun_out/${tag}_pmc/k3_hbm_traffic_headline.json
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
for k in ("value","ms_per_step","match_wall_ms","match_pairs_per_s","match_stages_ms","latency","kernel_ms_per_step","parity_check"):
    print(k, d.get(k))
print({k:d["roofline"].get(k) for k in ("kernel","achieved","peak","frac","frac_hbm_priced","frac_lds_floor","lds_floor_ms","traffic","compulsory_bytes","avg_launch_ms","symmetric_form","multiply_adds_executed")})
for name, c in d.get("configs", {}).items():
    print("==", name, {k: c.get(k) for k in ("error","ms_per_step","match_wall_ms","kernel_ms_per_step","bench_wall_s")})
    print("   roofline", {k: (c.get("roofline") or {}).get(k) for k in ("kernel","achieved","peak","frac","frac_lds_floor","traffic","unit","scored_fraction")})
    print("   cpu", (c.get("cpu_baseline") or {}).get("value"), "parity", (c.get("parity_check") or {}).get("ok", (c.get("parity_check") or {}).get("bit_exact")))
print(json.load(open("gpurun_out/${tag}_pmc/k3_hbm_traffic_headline.json"))["records"].get("headline"))
PY
head -32 gpurun_out/${tag}_profile/summary_headline.txt
