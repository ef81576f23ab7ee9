#!/bin/bash
# one-off K3 experiments: each line is a set of env knobs; prints K3 ms per step
run() {
  echo -n "$*  ->  "
  env "$@" python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('k3_ms', k['k3_cossim_topn'], 'step_ms', round(d['ms_per_step'],3))"
}
while read -r line; do
  [ -z "$line" ] && continue
  run $line
done < "${1:-/dev/stdin}"
