#!/bin/bash
# SQ counters of the K4 kernels on config 3 (real IMDB 20k x 20k)
R=$PWD
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES"
P3="SQ_WAVES SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $P --kernel-trace -d $R/gpurun_out/k4_pmc/p$i -o k4 --output-format csv -- python $R/bench.py --config editdistance --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/k4_pmc_p$i.log 2>&1
done
cd $R && python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/k4_pmc/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        if 'k4_indel' in k: print(k, n[(k, list(d)[0])], {c: round(v / max(1, n[(k, c)])) for c, v in d.items()})
PY
tail -2 gpurun_out/k4_pmc_p1.log
