#!/bin/bash
# End-of-round evidence of round 6 in ONE gpurun call, from the build this snapshot holds: the whole GPU suite + smoke, the default
# bench line, rocprofv3 stats + PMC passes of the same command, K3's fabric traffic on the three TF-IDF workloads, K7's timings.
# Run it through tools/r6_final_local.sh (build container): that wrapper stamps the commit the sources are at (.git does not
# travel), refuses a dirty tree, and files the results under profiles/ only if HEAD is still that commit when the call returns.
# usage (GPU box): bash tools/r6_final.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
HEAD_ID=$(cat tools/.build_head 2>/dev/null) || { echo "tools/.build_head is missing: run tools/r6_final_local.sh"; exit 2; }
O=gpurun_out/r06_final; rm -rf $O; mkdir -p $O; echo "$HEAD_ID" > $O/source_commit.txt
timeout 2400 python -m pytest tests/ -m gpu -q --timeout 900 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
timeout 1200 bash tools/profile_bench.sh $O/profile > $O/profile.log 2>&1
timeout 900 bash tools/pmc_traffic.sh $O/pmc > $O/pmc.log 2>&1
timeout 600 bash tools/profile_other.sh $O/other > $O/other.log 2>&1
timeout 300 python tools/k7_time.py 20000 WRatio,partial_ratio,token_ratio,partial_token_ratio > $O/k7_fuzz.txt 2>&1
timeout 300 python tools/k7_rowstats.py WRatio >> $O/k7_fuzz.txt 2>&1
# the user-level call: round 5's form (a session launch per range) against the streamed session, then the host side on one thread
# (round 6's first form) against the crews of host threads (packer + frame), one process, one box
timeout 300 python tools/r6_match_ab.py "round 5 (launch per range, copying upload, one host thread):PFZ_K3_NO_STREAMED=1,PFZ_MATCH_SHARES=0.3;0.3;0.25;0.15,PFZ_DIRECT_PACK=0,PFZ_HOST_THREADS=1" "launch per range, direct pack, one host thread:PFZ_K3_NO_STREAMED=1,PFZ_MATCH_SHARES=0.3;0.3;0.25;0.15,PFZ_HOST_THREADS=1" "streamed12, copying upload, one host thread:PFZ_DIRECT_PACK=0,PFZ_HOST_THREADS=1,PFZ_RANGE_FILL=0" "streamed5, one host thread:PFZ_MATCH_SHARES=1;1;1;1;1,PFZ_HOST_THREADS=1,PFZ_RANGE_FILL=0" "streamed12, one host thread (round 6, first form):PFZ_HOST_THREADS=1,PFZ_RANGE_FILL=0" "streamed12, range fill on one thread:PFZ_HOST_THREADS=1" "streamed12, crews of 2:PFZ_HOST_THREADS=2" "streamed12, crews of 4:PFZ_HOST_THREADS=4" "streamed12, crews of 8 (default):" "streamed12, packer alone on 8:PFZ_RANGE_THREADS=1" "streamed12, frame alone on 8:PFZ_PACK_INTO_THREADS=1" "streamed12, crews of 8, confinement lifted (the pool's threads stay where they last ran -- threads the scheduler places afresh: frame_fill_mt.txt):PFZ_HOST_PIN=0" "streamed16, crews of 8:PFZ_MATCH_SHARES=1;1;1;1;1;1;1;1;1;1;1;1;1;1;1;1" > $O/match_ab.txt 2>&1
# what the frame fill's variants cost on this host's CPU, on threads the scheduler places and on threads on the caller's L3
(gcc -O3 -msse4.1 tools/ubench/frame_fill_mt.c -lpthread -lm -o /tmp/ffmt && /tmp/ffmt && /tmp/ffmt pin) > $O/frame_fill_mt.txt 2>&1
# the streamed session's range chains on 1 / 2 (shipped) / 3 / 4 side streams: variants/sidesK.so, built by tools/r6_final_local.sh from this commit
timeout 600 bash tools/r6_sides_ab.sh > $O/side_streams.txt 2>&1
# `python bench.py --gpus 2` as the driver would call it, on a box with one device: launches itself, every rank says what is missing
timeout 300 python bench.py --gpus 2 --steps 2 > $O/gpus2_selflaunch.txt 2>&1; echo "bench --gpus 2 rc=$? (expected: not 0 on a one-GPU box)" >> $O/gpus2_selflaunch.txt
for f in $O/profile/summary.txt $O/profile/summary_headline.txt $O/profile/summary_match.txt $O/k7_fuzz.txt $O/match_ab.txt $O/frame_fill_mt.txt $O/side_streams.txt; do [ -f $f ] && sed -i "1i (source commit $HEAD_ID; tools/r6_final.sh)" $f; done
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k: d.get(k) for k in ("value","ms_per_step","match_wall_ms","latency_top1_ms","kernel_ms_per_step","device_step","k3_ms_per_step_inside_the_call")})
print({k:d["roofline"].get(k) for k in ("bound","frac","frac_hbm_priced","lds_floor_ms","avg_launch_ms","traffic","symmetric_form")})
print("parity", (d.get("parity_check") or {}).get("ok"), "pins", d.get("library_pins"))
for name, c in d.get("configs", {}).items():
    print("==", name, {k: c.get(k) for k in ("error","ms_per_step","match_wall_ms")}, (c.get("roofline") or {}).get("frac"), (c.get("parity_check") or {}).get("ok", (c.get("parity_check") or {}).get("bit_exact")))
PY
head -24 $O/profile/summary_headline.txt
