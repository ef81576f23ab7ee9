cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --no-configs --steps 12 --warmup 3 --no-cpu-baseline --no-match-wall"
for v in 18 36 72 144 400; do echo -n "[per CU $v] "; PFZ_K3_SYM_P0_PER_CU=$v timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', round(d['ms_per_step'],4), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'])"; done
