#!/bin/bash
# round 5: what pass 0 of K3's symmetric form spends its time on: what-if variants (variants/p0_exp*.so: -DPFZ_K3_SYM_EXP=10 no warm start,
# 11 no final compaction, 12 no sweep, 13 no scatter; results wrong on purpose), per-pass kernel times.  usage (GPU box): bash tools/r5_k3_p0.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_k3_p0; rm -rf $O; mkdir -p $O
B="python bench.py --no-configs --steps 10 --warmup 2 --no-cpu-baseline --no-match-wall"
for v in "" variants/p0_exp*.so; do
  n=$(basename "${v:-product}" .so)
  if [ -n "$v" ]; then export POLYFUZZ_HIP_LIB=$PWD/$v; else unset POLYFUZZ_HIP_LIB; fi
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/$n -o bench -- $B > $O/$n.log 2>&1
  python tools/rocprof_summary.py $O/$n/bench_results.db 2>&1 | grep "k3_sym_kernel<2048, [01]>" | cut -c1-110 | sed "s/^/$n /"
done
