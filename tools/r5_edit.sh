#!/bin/bash
# round 5: K4 + K7 after the bitop3 / preload changes: A/B on one box, their GPU tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/r5_k4.sh
timeout 1200 python -m pytest tests/test_indel_gpu.py tests/test_fuzz_gpu.py tests/test_matchers_gpu.py tests/test_knobs_gpu.py -m gpu -q --timeout 600 2>&1 | tail -3
