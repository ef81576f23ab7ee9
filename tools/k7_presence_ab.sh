#!/bin/bash
# First GPU call of the next round: the symbol-presence build of K7 ("What comes next" 2) against the default --
# its parity tests, then both libraries side by side on this box, then the hand-over thresholds again for the variant
# (fewer pairs survive per row).  Run ON THE GPU BOX after building the variant HERE (it travels with the snapshot):
#   mkdir -p variants && bash tools/build_variant.sh $PWD/variants/k7_pres.so -DPFZ_K7_PRESENCE
#   cp polyfuzz_amd/libpolyfuzz_hip.so variants/k7_default.so
#   gpurun --timeout 900 -- 'bash tools/k7_presence_ab.sh'
mkdir -p gpurun_out
POLYFUZZ_HIP_LIB=$PWD/variants/k7_pres.so timeout 400 python -m pytest tests/test_fuzz_gpu.py tests/test_facade_flow_gpu.py -m gpu -q -x --timeout 200 > gpurun_out/k7_pres_tests.log 2>&1
echo "presence build, tests rc=$?"; tail -3 gpurun_out/k7_pres_tests.log
bash tools/k7_ab.sh k7_pres variants/k7_default.so variants/k7_pres.so | sed 's/scored/\n    scored/'
bash tools/k7_hand_sweep.sh variants/k7_pres.so
