#!/bin/bash
# dynamic instruction counts of K3 per part: full kernel, sweep + top-n only (PFZ_K3_ABLATE=1), scatter only (=2)
OUT=${1:-gpurun_out/k3_split}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
for A in 0 1 2; do
  PFZ_K3_ABLATE=$A timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace \
     -d "$OUT/a$A" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-match-wall > "$OUT/a$A.log" 2>&1
  echo "== ablate=$A"
  python tools/rocprof_summary.py "$OUT/a$A/bench_results.db" | grep "k3_cossim" | grep -v "void pfz::k3_cossim_topn_kernel<2048, 96>(int const\*, int co"
done
