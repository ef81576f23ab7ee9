#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of K3 for the three TF-IDF workloads of the bench (separate PMC passes, kernel-trace only -- the
# combination gpurun allows), then profiles-ready JSON.  Run on the GPU box: bash tools/pmc_traffic.sh [outdir]
OUT=${1:-gpurun_out/pmc_traffic}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
for cfg in tfidf c2 tfidf_1m; do
  extra=""; [ $cfg = tfidf ] && extra="--no-configs --step device"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --pmc $c --kernel-trace -d "$OUT/${cfg}_$c" -o bench -- python bench.py --config $cfg --steps 2 --warmup 1 \
      --no-cpu-baseline --no-match-wall $extra > "$OUT/${cfg}_$c.log" 2>&1
    echo "$cfg $c rc=$?"
  done
done
python tools/pmc_traffic.py "$OUT" > "$OUT/k3_hbm_traffic.json" && cat "$OUT/k3_hbm_traffic.json" | head -60
