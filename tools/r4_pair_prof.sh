#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
PFZ_K3_PAIR=1 timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/pair_prof -o p -- python bench.py --no-cpu-baseline --no-match-wall --no-configs --steps 10 > gpurun_out/pair_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/pair_prof/p_results.db | head -12
