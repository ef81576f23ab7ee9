#!/bin/bash
# SQ counters of K7 (WRatio, IMDB 20k x 20k).  usage: bash tools/k7_pmc.sh [env assignments, e.g. PFZ_K7_EXP=1]
R=$PWD
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES"
i=0
rm -rf $R/gpurun_out/k7_pmc
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $P --kernel-trace -d $R/gpurun_out/k7_pmc/p$i -o k7 --output-format csv -- python $R/tools/k7_time.py 20000 WRatio > $R/gpurun_out/k7_pmc_p$i.log 2>&1
done
cd $R && python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/k7_pmc/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        if 'k7_fuzz_kernel' in k: print(k, n[(k, list(d)[0])], {c: round(v / max(1, n[(k, c)])) for c, v in d.items()})
PY
