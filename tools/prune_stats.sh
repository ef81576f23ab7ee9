#!/bin/bash
# work counters of K3 (PFZ_K3_STATS) for a few pruning settings, one bench step each
for V in "32 50" "32 75" "64 75" "64 100"; do
  set -- $V
  echo "heavy=$1 alpha=$2"
  PFZ_K3_STATS=1 PFZ_K3_HEAVY=$1 PFZ_K3_ALPHA=$2 python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "k3 stats" | tail -1
done
