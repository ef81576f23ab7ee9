#!/bin/bash
# PMC passes over bench.py for the K3 kernel (run on the GPU box via gpurun).  Extra env (PFZ_K3_*) is inherited.
# usage: tools/pmc_k3.sh <outdir>
OUT=${1:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
run() { rocprofv3 --pmc $2 --kernel-trace -d "$OUT/$1" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/$1.log" 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
run sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
run sq3 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"
run ta "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python tools/rocprof_summary.py $OUT/sq1/bench_results.db $OUT/sq2/bench_results.db $OUT/sq3/bench_results.db $OUT/ta/bench_results.db $OUT/tcc/bench_results.db $OUT/fetch/bench_results.db $OUT/write/bench_results.db | grep "k3_cossim\|==" | cut -c1-120
