for v in "" "PFZ_HOST_PIN=0" "PFZ_HOST_THREADS=2" "PFZ_HOST_THREADS=8" "PFZ_RANGE_FILL=0 PFZ_HOST_THREADS=1" ""; do
  echo "== $v"
  env $v python bench.py --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'],3), 'match_wall_ms', d.get('match_wall_ms'), 'device', round(d['device_step']['ms_per_step'],3))"
done
