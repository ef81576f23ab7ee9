#!/bin/bash
# the whole GPU suite + smoke (what the driver runs at round end), then a probe of the string packer's thread count
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/ -m gpu -q --timeout 600 > gpurun_out/r4_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r4_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for t in 1 4 8 16; do
  echo "PFZ_PACK_THREADS=$t: $(PFZ_PACK_THREADS=$t timeout 100 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from polyfuzz_amd import datasets
from polyfuzz_amd.models import TFIDF
names = datasets.load_company_names()
m = TFIDF(min_similarity=0, top_n=5)
for _ in range(3): m.match(names)
ts, st = [], []
for _ in range(9):
    df = None
    t0 = time.perf_counter(); df = m.match(names); ts.append((time.perf_counter() - t0) * 1e3); st.append(m.last_timings)
i = int(np.argsort(ts)[4]); print(f"wall {ts[i]:.2f} ms", {k: round(v, 2) for k, v in st[i].items()})
PY
)"
done
