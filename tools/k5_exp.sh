#!/bin/bash
# what-if builds of the K5 GEMM (tools/build_variant.sh ... -DPFZ_K5_EXP=n -> polyfuzz_amd/_k5exp<n>.so), top-n not overlapped
export K5_FILLS=1 PFZ_K5_NO_OVERLAP=1
echo "== as shipped, row top-n overlapped"; env -u PFZ_K5_NO_OVERLAP python tools/k5_gemm_time.py
echo "== as shipped, serial"; python tools/k5_gemm_time.py
for e in 2 3 4; do
  [ -f polyfuzz_amd/_k5exp$e.so ] && { echo "== EXP $e"; POLYFUZZ_HIP_LIB=$PWD/polyfuzz_amd/_k5exp$e.so python tools/k5_gemm_time.py; }
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 -d $GRAFT_REPO_ROOT/gpurun_out/k5_pmc -o k5 --output-format csv -- python $GRAFT_REPO_ROOT/tools/k5_gemm_time.py 16384 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/k5_pmc/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        if 'gemm' in k: print(k, {c: v / max(1, n[(k, c)]) for c, v in d.items()})
PY
