lscpu | grep -E "Model name|Socket|NUMA|Thread|Core|CPU\(s\)" | head -20
for f in /sys/class/drm/card*/device/numa_node; do echo $f $(cat $f); done 2>/dev/null | head
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null
python - <<'PY'
import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu now', (getattr(os, "sched_getcpu", lambda: -1)()))
PY
cd $GRAFT_REPO_ROOT
for c in 0 8 64 128 192 255; do echo "== taskset -c $c"; taskset -c $c python tools/r6_match_ab.py "x:" 2>&1 | tail -1 | cut -c45-200; done
echo "== unpinned x3"; for i in 1 2 3; do python tools/r6_match_ab.py "x:" 2>&1 | tail -1 | cut -c45-200; done
