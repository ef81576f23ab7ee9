#!/bin/bash
# round 5: K3 headline timing under a list of environment settings ("A=1 B=2" per argument).  usage (GPU box): bash tools/r5_k3_env.sh "" "PFZ_X=1" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --no-configs --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall"
for e in "$@"; do env $e timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('  [$e]: step', round(d['ms_per_step'],4), 'k3', k['k3_cossim_topn'])"; done
