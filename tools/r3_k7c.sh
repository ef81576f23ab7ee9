#!/bin/bash
# where K7's wave time goes.  usage: bash tools/r3_k7c.sh <tag> [scorer]
tag=${1:-k7c}
mkdir -p gpurun_out
timeout 300 python tools/k7_rowstats.py ${2:-WRatio} > gpurun_out/${tag}_rowstats.log 2>&1
cat gpurun_out/${tag}_rowstats.log | tail -40
