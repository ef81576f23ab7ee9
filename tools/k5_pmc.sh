#!/bin/bash
# SQ counters of the K5 GEMM (16384^2 x 768)
export K5_FILLS=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -i "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u | tr '\n' ' '; echo
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"
P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $P --kernel-trace -d $R/gpurun_out/k5_pmc/p$i -o k5 --output-format csv -- python $R/tools/k5_gemm_time.py 16384 > $R/gpurun_out/k5_pmc_p$i.log 2>&1
done
cd $R && python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/k5_pmc/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:28]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        if 'gemm' in k: print(k, {c: round(v / max(1, n[(k, c)])) for c, v in d.items()})
PY
tail -3 gpurun_out/k5_pmc_p2.log
