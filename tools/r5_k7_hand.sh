#!/bin/bash
# round 5: K7's hand-over constants swept again on the faster kernel (WRatio, 20k x 20k): PFZ_K7_HAND=batches,min groups,units,short len,short batches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for h in "" "48,16,64,8,24" "64,16,64,8,16" "64,16,64,12,32" "96,16,64,8,32" "64,8,64,8,32" "64,16,32,8,32" "64,16,128,8,32" "40,16,64,10,20"; do
  if [ -n "$h" ]; then export PFZ_K7_HAND=$h; else unset PFZ_K7_HAND; fi
  python tools/k7_time.py 20000 WRatio 2>&1 | grep "20000 x" | cut -c28-100 | sed "s|^|[${h:-default}] |"
done; done
