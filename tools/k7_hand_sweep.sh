#!/bin/bash
# hand-over tuning on one box: PFZ_K7_HAND=<batches of surviving pairs that make a row heavy>,<min groups>,<most continuation
# units>,<from-strings up to this long ...>,<... are heavy from this many batches on>
lib=${1:-polyfuzz_amd/libpolyfuzz_hip.so}
mkdir -p gpurun_out
for h in 64,16,64,0,64 64,16,64,8,32 64,16,64,8,24 64,16,64,8,16 64,16,64,6,24 64,16,64,10,32 64,16,64,12,40 64,16,64,8,48; do
  echo "HAND=$h: $(PFZ_K7_HAND=$h POLYFUZZ_HIP_LIB=$lib timeout 100 python tools/k7_time.py 20000 WRatio,partial_ratio,partial_token_ratio 2>&1 | grep '20000 x' | sed 's/k7_prepare.*//' | sed 's/step.*k7_fuzz//' | sed 's/20000 x 20000: //' | tr '\n' ' ')"
done 2>&1 | tee gpurun_out/k7_hand_sweep.log
