#!/bin/bash
# Build container side of tools/r6_final.sh: the evidence under profiles/r06_* comes from ONE commit, and says which.
#   1. the tree must be clean (what is profiled is what is committed);  2. the library is rebuilt from it;  3. HEAD's id travels to
#   the GPU box in tools/.build_head (git-ignored: .git does not travel);  4. after the call HEAD must still be that commit --
#   otherwise nothing is filed;  5. the summaries are copied to profiles/ (a follow-up commit that touches only profiles/ and docs).
set -e
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain -- . ':!profiles' ':!DESIGN.md' ':!README.md' ':!tools/README.md' ':!INTEGRATION.md')" ]; then echo "working tree is not clean: commit first"; git status --short | head; exit 1; fi
HEAD_ID=$(git rev-parse HEAD); echo "$HEAD_ID" > tools/.build_head
python -c "import __graft_entry__ as g; g.build()" | tail -1
mkdir -p variants; for k in 1 3 4; do bash tools/build_variant.sh "$PWD/variants/sides$k.so" -DPFZ_K3_SYM_SIDE_STREAMS=$k 2>&1 | grep -i " error" || true; done
gpurun --timeout 5400 -- 'bash tools/r6_final.sh' 2>&1 | tail -60
[ "$(git rev-parse HEAD)" = "$HEAD_ID" ] || { echo "HEAD moved while the call ran: results NOT filed"; exit 1; }
[ "$(cat gpurun_out/r06_final/source_commit.txt)" = "$HEAD_ID" ] || { echo "the box profiled another build: results NOT filed"; exit 1; }
O=gpurun_out/r06_final
cp $O/bench.json profiles/r06_bench.json
cp $O/profile/summary.txt profiles/r06_bench_rocprofv3_summary.txt
cp $O/profile/summary_headline.txt profiles/r06_bench_rocprofv3_summary_headline.txt
cp $O/profile/summary_match.txt profiles/r06_bench_rocprofv3_summary_match_step.txt
cp $O/gpu_tests.log profiles/r06_gpu_tests.log; cp $O/smoke.log profiles/r06_smoke.log
cp $O/match_ab.txt profiles/experiments/r06_match_streamed_ab.txt; grep "bench rank\|bench\]\|rc=" $O/gpus2_selflaunch.txt > profiles/r06_bench_gpus2_selflaunch.txt
cp $O/profile/bench_under_rocprof.json profiles/r06_bench_under_rocprof.json 2>/dev/null || true
cp $O/k7_fuzz.txt profiles/r06_k7_fuzz.txt
cp $O/frame_fill_mt.txt profiles/experiments/r06_frame_fill_mt.txt
cp $O/side_streams.txt profiles/experiments/r06_k3_side_streams.txt
[ -f $O/other/summary_other.txt ] && cp $O/other/summary_other.txt profiles/r06_other_kernels_rocprofv3_summary.txt
python - <<PY
import json
new = json.load(open("$O/pmc/k3_hbm_traffic.json"))
for r in new["records"].values():
    r["source_commit"] = "$HEAD_ID"
json.dump(new, open("profiles/k3_hbm_traffic.json", "w"), indent=1)
print("traffic records:", {k: round(v["hbm_bytes_per_launch"] / 1e9, 3) for k, v in new["records"].items()}, "GB per launch")
PY
echo "filed under profiles/r06_* (source commit $HEAD_ID)"
