#!/bin/bash
# round 5: the short-list paths after a change: vectoriser / matcher / K3 tests, config 2 and the headline (three runs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_vectorize_gpu.py tests/test_matchers_gpu.py tests/test_random_parity_gpu.py tests/test_knobs_gpu.py tests/test_facade_flow_gpu.py tests/test_k3_cossim_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -3
for rep in 1 2 3; do python bench.py --config c2 --steps 30 --warmup 5 --no-cpu-baseline --no-match-wall 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 ms_per_step %.4f' % r['ms_per_step'])"; done
python bench.py --no-configs --steps 12 --warmup 3 --no-cpu-baseline --no-match-wall 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline ms_per_step %.4f' % r['ms_per_step'])"
