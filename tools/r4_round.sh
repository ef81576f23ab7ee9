#!/bin/bash
# End-of-round evidence of round 4 in one gpurun call: smoke, the default bench line, rocprofv3 stats + PMC passes of the same
# command, the FETCH / WRITE traffic of K3 on the three TF-IDF workloads, K7's timings.  (GPU tests: separate call.)
tag=${1:-r04}
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; tail -c 400 gpurun_out/${tag}_bench.err
timeout 300 python tools/k7_time.py 20000 WRatio,partial_ratio,token_ratio,partial_token_ratio names > gpurun_out/${tag}_k7.log 2>&1
timeout 200 python tools/k7_rowstats.py WRatio 2>&1 | grep -E "as shipped|wave time|timeline|scoring batches|began" >> gpurun_out/${tag}_k7.log
timeout 900 bash tools/profile_bench.sh gpurun_out/${tag}_profile > gpurun_out/${tag}_profile.log 2>&1
timeout 600 bash tools/pmc_traffic.sh gpurun_out/${tag}_pmc_traffic > gpurun_out/${tag}_pmc_traffic.log 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
for k in ("value","ms_per_step","match_wall_ms","match_pairs_per_s","latency","kernel_ms_per_step","parity_check"):
    print(k, d.get(k))
print({k:d["roofline"].get(k) for k in ("achieved","peak","frac","frac_hbm_priced","frac_lds_floor","traffic","compulsory_bytes","avg_launch_ms")})
for name, c in d.get("configs", {}).items():
    print("==", name, {k: c.get(k) for k in ("error","ms_per_step","match_wall_ms","kernel_ms_per_step","bench_wall_s")})
    print("   roofline", {k: (c.get("roofline") or {}).get(k) for k in ("kernel","achieved","peak","frac","frac_lds_floor","traffic","unit","scored_fraction")})
    print("   cpu", (c.get("cpu_baseline") or {}).get("value"), "parity", (c.get("parity_check") or {}).get("ok", (c.get("parity_check") or {}).get("bit_exact")))
PY
tail -8 gpurun_out/${tag}_k7.log
head -30 gpurun_out/${tag}_profile/summary_headline.txt
grep -A12 '"tfidf_1m_shard"' gpurun_out/${tag}_pmc_traffic/k3_hbm_traffic.json | grep -E "kernel|hbm_bytes"
