#!/bin/bash
# One gpurun call of the build -> measure loop: GPU tests, the default bench line, frame-thread sweep.
# usage (on the GPU box, from the repo root): bash tools/gpu_round.sh <tag>
tag=${1:-run}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
tail -5 gpurun_out/${tag}_tests.log
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/${tag}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
for k in ("value","ms_per_step","match_wall_ms","match_wall_ms_min","match_stages_ms","latency","kernel_ms_per_step","parity_check","cpu_baseline_arms"):
    print(k, d.get(k))
print({k:d["roofline"][k] for k in ("achieved","frac","avg_launch_ms","lds_floor_ms","frac_of_lds_floor")})
PY
for t in 1 2 4 8; do
  PFZ_FRAME_THREADS=$t python - <<PY
import sys, time; sys.path.insert(0,'.')
import numpy as np
from polyfuzz_amd import datasets
from polyfuzz_amd.models import TFIDF, _utils
names=datasets.load_company_names()
m=TFIDF(min_similarity=0,top_n=5)
for _ in range(3): m.match(names)
ts=[]; st=[]
for _ in range(9):
    t0=time.perf_counter(); m.match(names); ts.append((time.perf_counter()-t0)*1e3); st.append(m.last_timings)
i=int(np.argsort(ts)[len(ts)//2])
print("frame threads", _utils._FILL_THREADS, "match ms median %.2f min %.2f"%(ts[i],min(ts)), {k:round(v,2) for k,v in st[i].items()})
PY
done
nproc; lscpu | grep "Model name"
