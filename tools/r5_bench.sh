#!/bin/bash
# round 5: the driver's default bench line (all configs) -> gpurun_out/r5_bench_<tag>.json + a short digest.  usage (GPU box): bash tools/r5_bench.sh [tag] [bench flags]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; T=${1:-a}; shift; mkdir -p gpurun_out
timeout 900 python bench.py "$@" > gpurun_out/r5_bench_$T.json 2> gpurun_out/r5_bench_$T.err; echo "bench rc=$?"; tail -2 gpurun_out/r5_bench_$T.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r5_bench_$T.json').read().strip().splitlines()[-1])
print('value %.4g %s  ms_per_step %.4f  match_wall_ms %s  latency_top1_ms %s'%(d['value'],d['unit'],d['ms_per_step'],d.get('match_wall_ms'),d.get('latency_top1_ms')))
print('kernel_ms', d.get('kernel_ms_per_step'))
rf=d.get('roofline',{}); print('roofline', {k:rf.get(k) for k in ('bound','frac','frac_hbm_priced','lds_floor_ms','avg_launch_ms','symmetric_form')})
print('stages', d.get('match_stages_ms')); print('parity', {k:v for k,v in (d.get('parity_check') or {}).items() if k in ('ok','rows_checked','max_abs_score_err','index_diffs_not_near_ties')}, (d.get('parity_check') or {}).get('vectoriser',{}).get('ok'))
print('cpu', (d.get('cpu_baseline') or {}).get('value'))
for k,c in (d.get('configs') or {}).items():
    rf=c.get('roofline') or {}
    print(' ',k,'ms_per_step %.4f'%c['ms_per_step'],'match_wall',c.get('match_wall_ms'),'frac',rf.get('frac'),'kernel',c.get('kernel_ms_per_step'),'parity',(c.get('parity_check') or {}).get('ok',(c.get('parity_check') or {}).get('bit_exact')))
PY
