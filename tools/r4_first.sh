#!/bin/bash
# round 4, first GPU call: (1) the three micro-benchmarks whose constants bench.py's rooflines use, outputs tracked under
# profiles/; (2) the symbol-presence build of K7 (tests, same-box A/B, hand-over sweep).
mkdir -p gpurun_out/r04_ubench
for b in lds_atomic valu_rate fetch_calib; do
  timeout 120 tools/ubench/$b.bin > gpurun_out/r04_ubench/$b.txt 2>&1; echo "$b rc=$?"
done
cat gpurun_out/r04_ubench/lds_atomic.txt gpurun_out/r04_ubench/valu_rate.txt
bash tools/k7_presence_ab.sh
