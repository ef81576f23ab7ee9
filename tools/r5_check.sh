#!/bin/bash
# the whole GPU suite + smoke (what the driver runs at round end)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -m gpu -q --timeout 900 -x > gpurun_out/r5_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/r5_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
