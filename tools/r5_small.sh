#!/bin/bash
# round 5: the short-list paths (fused launches of a transform): tests, the single-query timeline, config 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vectorize_gpu.py tests/test_matchers_gpu.py tests/test_facade_flow_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -4
bash tools/r5_query_trace.sh 2>&1 | grep -v "^W2026\|^E2026" | tail -12
for i in 1 2; do python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-match-wall 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 ms_per_step', r['ms_per_step'])"; done
