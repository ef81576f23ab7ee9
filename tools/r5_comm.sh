#!/bin/bash
# round 5: the multi-rank paths on one device (local transport) + bench's rank logic.  usage (GPU box): bash tools/r5_comm.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_comm; mkdir -p $O
timeout 1200 python -m pytest tests/test_comm_gpu.py tests/test_bench_local_gpu.py -m gpu -q -x --timeout 900 > $O/tests.log 2>&1
echo "tests rc=$?"; tail -15 $O/tests.log
