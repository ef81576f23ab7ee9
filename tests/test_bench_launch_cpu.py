"""`python bench.py --gpus N` as the DRIVER calls it -- no torch.distributed.run in front, no WORLD_SIZE in the environment
(that call used to end in a SystemExit telling the caller to use the launcher).  bench.py now becomes the
launcher itself: one rank per GPU, rendezvous over gloo (torch.distributed is never an RCCL user), and
* on a box without enough devices every rank says so -- "N devices needed, M visible" -- and the job's exit code is not 0;
* `--rehearse-cpu` (tests only: bench.py hands its ranks to tests/bench_rehearsal.py) drives the rank logic behind the launcher --
  barrier / max-over-ranks around the timed steps, `TfidfMatchJob`'s sharded self-match with its collective questions and
  exchanges -- through tests/cpu_engine.py at world 2, and the result is the single-process oracle's."""
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

pytest.importorskip("torch")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)


def test_gpus_2_launches_itself_and_reports_missing_devices():
    from polyfuzz_amd import _lib
    if _lib.device_count() >= 2:
        pytest.skip("this box has two devices: the launch would run the real bench")
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0
    assert "must be launched with torch.distributed.run" not in r.stderr + r.stdout
    for rank in (0, 1):
        assert f"[bench rank {rank}] rendezvous ok (gloo, world 2)" in r.stderr
        assert f"[bench rank {rank}] --gpus 2: 2 devices needed, {_lib.device_count()} visible" in r.stderr
    assert "librccl" in r.stderr and "compiled against rccl.h" in r.stderr
    assert not r.stdout.strip().startswith("{")            # no bench line from a job that did not run


def test_rank_logic_behind_the_launcher_at_world_2(oracle_mod):
    r = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse-cpu")
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1                                   # rank 0 alone prints
    rec = json.loads(line[0])
    assert rec["world"] == 2 and rec["result_is_full"] and "value" not in rec and "metric" not in rec
    from polyfuzz_amd import synth
    names = synth.company_names(400, seed=3)
    v = oracle_mod.TfidfOracle().fit(names)                 # reference _tfidf.py:113-116
    a3 = v.transform(names)
    e_idx, e_val = oracle_mod.cossim_topn(a3, a3, len(v.vocabulary), 3, 0.0, exclude_diag=True)
    assert rec["idx_crc32"] == zlib.crc32(np.ascontiguousarray(e_idx, np.int32).tobytes())
    assert rec["val_crc32"] == zlib.crc32(np.ascontiguousarray(e_val, np.float64).tobytes())
