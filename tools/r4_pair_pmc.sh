#!/bin/bash
OUT=gpurun_out/pair_pmc
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
run() { PFZ_K3_PAIR=$3 timeout 200 rocprofv3 --pmc $2 --kernel-trace -d "$OUT/$1" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-match-wall --no-configs > "$OUT/$1.log" 2>&1; }
for p in 1 0; do
run sq1_$p "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" $p
run sq2_$p "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" $p
done
python tools/rocprof_summary.py $OUT/sq1_1/bench_results.db $OUT/sq2_1/bench_results.db $OUT/sq1_0/bench_results.db $OUT/sq2_0/bench_results.db | grep "k3_\|==" | cut -c1-110
PFZ_K3_PAIR=1 PFZ_K3_PAIR_DEBUG=1 timeout 100 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-match-wall --no-configs 2>&1 | grep "k3_pair:" | head -2
