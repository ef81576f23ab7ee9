#!/bin/bash
# round 5: SQ counters of K3's symmetric passes (instruction mix, LDS, instruction cache).  usage (GPU box): bash tools/r5_k3_pmc.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; T=${1:-a}; O=gpurun_out/r5_k3_pmc_$T; mkdir -p $O
B="python bench.py --no-configs --steps 3 --warmup 1 --no-cpu-baseline --no-match-wall"
run() { timeout 200 rocprofv3 --pmc $2 --kernel-trace -d "$O/$1" -o bench -- $B > "$O/$1.log" 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
run sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
run sqc "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQC_DCACHE_REQ SQC_DCACHE_MISSES"
python tools/rocprof_summary.py $O/sq1/bench_results.db $O/sq2/bench_results.db $O/sqc/bench_results.db | grep "k3_sym_kernel\|==" | cut -c1-120 | tee $O/summary.txt
tail -3 $O/sqc.log
