#!/bin/bash
# first GPU run of K3's symmetric form (k3_symmetric.hip): its tests, the headline with and without it (parity leg on),
# the split of TFIDF.match() over row ranges.  usage (GPU box): bash tools/r4_sym.sh
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_sym; mkdir -p $O
timeout 400 python -m pytest tests/test_k3_cossim_gpu.py -x -q -k "symmetric" > $O/tests.log 2>&1; echo "tests rc=$?"
tail -15 $O/tests.log
for sym in 0 auto; do
  if [ $sym = auto ]; then unset PFZ_K3_SYM; else export PFZ_K3_SYM=$sym; fi
  timeout 240 python bench.py --no-configs --steps 10 --warmup 2 --cpu-seconds 2 --no-match-wall > $O/bench_sym_$sym.json 2> $O/bench_sym_$sym.err; echo "bench sym=$sym rc=$?"
  python - $O/bench_sym_$sym.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("  ms_per_step", round(d["ms_per_step"], 4), "k3", d["kernel_ms_per_step"].get("k3_cossim_topn"), "sym", r.get("symmetric_form"),
          "frac_lds_floor", round(r["frac_lds_floor"], 3), "floor_ms", round(r["lds_floor_ms"], 3), "parity", d.get("parity_check", {}).get("ok"),
          d.get("parity_check", {}).get("rows_checked"), d.get("parity_check", {}).get("index_diffs_not_near_ties"), d.get("parity_check", {}).get("max_abs_score_err"))
except Exception as e:
    print("  no record:", e)
PY
done
unset PFZ_K3_SYM
timeout 200 python tools/match_split_probe.py > $O/match_split_sym.txt 2>&1; cat $O/match_split_sym.txt
PFZ_K3_SYM=0 timeout 60 python tools/match_split_probe.py 0.4,0.3,0.2,0.1 > $O/match_split_rowmajor.txt 2>&1; cat $O/match_split_rowmajor.txt
