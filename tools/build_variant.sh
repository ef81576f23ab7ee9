#!/bin/bash
# build a tuning variant of libpolyfuzz_hip.so: tools/build_variant.sh <out.so> -DPFZ_K3_SLOTS=16 ...
# use it with POLYFUZZ_HIP_LIB=<out.so>
OUT=$1; shift
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function "$@" \
  -I include -I polyfuzz_amd/csrc polyfuzz_amd/csrc/*.hip -L /opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o "$OUT"
