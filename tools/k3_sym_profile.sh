#!/bin/bash
# K3's symmetric form (csrc/k3_symmetric.hip): where its time goes.  Per-pass kernel times (rocprofv3 --kernel-trace --stats), what-if
# builds of the second filter (PFZ_K3_SYM_EXP: results wrong on purpose), one PMC pass.  usage (GPU box): bash tools/k3_sym_profile.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/k3_sym_profile; mkdir -p $O
B="python bench.py --no-configs --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- $B > $O/stats.log 2>&1; echo "stats rc=$?"
python tools/rocprof_summary.py $O/stats/bench_results.db 2>&1 | head -14 | cut -c1-130
for e in 0 1 3 4; do
  PFZ_K3_SYM_EXP=$e timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  exp $e: step', round(d['ms_per_step'],4), 'k3', d['kernel_ms_per_step']['k3_cossim_topn'])"
done
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS --kernel-trace -d $O/sq2 -o bench -- python bench.py --no-configs --steps 3 --warmup 1 --no-cpu-baseline --no-match-wall > $O/sq2.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace -d $O/sq1 -o bench -- python bench.py --no-configs --steps 3 --warmup 1 --no-cpu-baseline --no-match-wall > $O/sq1.log 2>&1
python tools/rocprof_summary.py $O/sq1/bench_results.db $O/sq2/bench_results.db 2>&1 | grep "k3_" | cut -c1-120
