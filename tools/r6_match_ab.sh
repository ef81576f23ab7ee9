#!/bin/bash
# round 6: TFIDF.match(names) on ONE box -- the round-5 form (a session launch per range, PFZ_K3_NO_STREAMED=1) against the streamed
# session (one pass-1 launch, ranges merged on a side stream), and pass 1's write-through stores against plain ones (variants/plain.so)
cd "$GRAFT_REPO_ROOT"
echo "== round-5 form: four session launches"; PFZ_K3_NO_STREAMED=1 python tools/match_split_probe.py 0.3,0.3,0.25,0.15
echo "== round-5 form, plain stores"; POLYFUZZ_HIP_LIB=variants/plain.so PFZ_K3_NO_STREAMED=1 python tools/match_split_probe.py 0.3,0.3,0.25,0.15
echo "== streamed"; python tools/match_split_probe.py 0.3,0.3,0.25,0.15 0.1,0.15,0.15,0.15,0.15,0.15,0.15 0.1,0.2,0.2,0.2,0.15,0.15
echo "== one launch, no ranges"; python tools/match_split_probe.py 1.0; POLYFUZZ_HIP_LIB=variants/plain.so python tools/match_split_probe.py 1.0
echo "== device step"; python bench.py --no-configs --no-cpu-baseline --no-match-wall | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])"
POLYFUZZ_HIP_LIB=variants/plain.so python bench.py --no-configs --no-cpu-baseline --no-match-wall | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain stores', d['ms_per_step'], d['kernel_ms_per_step'])"
