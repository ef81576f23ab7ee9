#!/bin/bash
# the 1M-shard configuration's bench record: vectoriser parity now runs on all 1 125 000 strings (oracle/tfidf_numpy.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout 900 python bench.py --config tfidf_1m --steps 3 --warmup 1 ) > gpurun_out/r5_1m.json 2> gpurun_out/r5_1m.err
echo "rc=$?"; tail -4 gpurun_out/r5_1m.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r5_1m.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"]); print(json.dumps(r.get("parity_check"), indent=1)[:2500])
PY
