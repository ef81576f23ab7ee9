"""bench.py's multi-rank logic rehearsed on ONE device: `--transport local --gpus 2` runs the same rank code as the
torch.distributed launch (sharding, barriers, max over ranks, the all-gather of the per-shard results through the
library's in-process transport) with two contexts and two host threads.  The JSON line is checked, and the gathered
result against the single-rank run of the same configuration."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *flags], capture_output=True, text=True, timeout=900,
                       cwd=REPO)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_tfidf_two_local_ranks(scaling):
    d = _bench("--transport", "local", "--gpus", "2", "--n", "20000", "--steps", "2", "--warmup", "1", "--scaling", scaling,
               "--no-cpu-baseline", "--no-match-wall")
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == scaling and d["config"]["transport"] == "local"
    assert "all-gather" in d["config"]["exchange"] and d["config"]["parallelism"].startswith("from-rows sharded x2")
    n_total = 40000 if scaling == "weak" else 20000
    # (strong: cost-balanced cuts of the sorted list, pipeline.balanced_bounds -- about half the rows each)
    assert d["config"]["n_from_total"] == n_total
    assert d["config"]["n_from_this_rank"] == 20000 if scaling == "weak" else 9000 < d["config"]["n_from_this_rank"] < 11000
    assert abs(d["value"] - n_total * 20000 * 2 / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["kernel"] == "k3_cossim_topn" and rf["bound"] == "hbm" and rf["frac"] > 0 and rf["frac"] == rf["frac_hbm_priced"]
    assert 0 < rf["frac_lds_floor"] <= 1.0 and rf["compulsory_bytes"] > 0 and rf["algorithmic_bytes_per_launch"] > rf["compulsory_bytes"]


@pytest.mark.parametrize("config", ["editdistance", "rapidfuzz", "dense"])
def test_other_configs_two_local_ranks(config):
    flags = ("--config", config, "--small", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1", "--no-match-wall")
    two = _bench("--transport", "local", "--gpus", "2", *flags)
    assert two["n_gpus"] == 2 and two["config"]["transport"] == "local" and "rehearsal" in two["config"]
    assert "sharded x2" in two["config"]["parallelism"] and "all-gather" in two["config"]["exchange"]
    chk = two["parity_check"]                     # the GATHERED result (rank 0's view of all rows) against the oracle
    assert chk.get("bit_exact", chk.get("ok")) is True, chk
    one = _bench(*flags)
    assert one["n_gpus"] == 1 and one["parity_check"].get("bit_exact", one["parity_check"].get("ok")) is True
    if config == "dense":
        assert two["scaling"] == "weak" and two["value"] > 0
    else:
        assert two["scaling"] == "strong" and two["config"]["n_from"] == one["config"]["n_from"] == 2000
