#!/bin/bash
# sixth GPU run: TFIDF.match() wall over ways of cutting the K3 launch, with the symmetric form
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_sym6; mkdir -p $O
timeout 300 python tools/match_split_probe.py 1.0 0.6,0.4 0.7,0.3 0.5,0.3,0.2 0.55,0.3,0.15 0.4,0.3,0.2,0.1 0.3,0.3,0.25,0.15 0.45,0.3,0.15,0.1 > $O/match_split.txt 2>&1; cat $O/match_split.txt
