#!/bin/bash
# occupancy the runtime reports for K7's kernels (PFZ_K7_DEBUG), per library variant
for so in "$@"; do echo "== $so"; PFZ_K7_DEBUG=1 POLYFUZZ_HIP_LIB=$so timeout 100 python tools/k7_time.py 20000 token_ratio 2>&1 | grep -E "k7 class|20000 x" | sort | uniq -c | head -8; done
