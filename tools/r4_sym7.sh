#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_sym7; mkdir -p $O
for e in 1 0 1 0; do echo "PFZ_EARLY_FRAME=$e"; PFZ_EARLY_FRAME=$e timeout 100 python tools/match_split_probe.py 0.3,0.3,0.25,0.15; done > $O/match_split_ab.txt 2>&1; cat $O/match_split_ab.txt
