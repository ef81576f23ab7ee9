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


@pytest.mark.parametrize("scaling", ["weak", "strong", None])
def test_tfidf_two_local_ranks(scaling):
    flags = ("--transport", "local", "--gpus", "2", "--n", "24000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-match-wall")
    d = _bench(*flags, *(("--scaling", scaling) if scaling else ()))
    scaling = scaling or "strong"                     # the default at N > 1: the one self-match cut over the ranks
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == scaling and d["config"]["transport"] == "local"
    assert "all-gather" in d["config"]["exchange"] and d["config"]["parallelism"].startswith("from-rows sharded x2")
    n_total = 48000 if scaling == "weak" else 24000
    assert d["config"]["n_from_total"] == n_total
    rf = d["roofline"]
    if scaling == "weak":
        # every rank: its own batch (the names in a rank-seeded order) against the replicated list -- the row-major kernel
        assert d["config"]["n_from_this_rank"] == 24000 and not rf["symmetric_form"]
    else:
        # K3's symmetric form cut over the ranks (24 000 rows: above the automatic choice's 20 480): rows r, r + 2, ...
        assert rf["symmetric_form"] and "candidate lists" in d["config"]["exchange"] and "once over all ranks" in d["config"]["parallelism"]
    assert abs(d["value"] - n_total * 24000 * 2 / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
    # the roofline's top level is the binding resource: a utilisation in (0, 1]; the HBM pricing sits beside it
    assert rf["kernel"].startswith("k3_cossim_topn") and rf["bound"] == "lds" and 0 < rf["frac"] <= 1.0 and rf["frac"] == rf["frac_lds_floor"]
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-9 and rf["frac_hbm_priced"] > 0 and rf["executed_bytes"] <= rf["algorithmic_bytes_per_launch"]
    assert rf["compulsory_bytes"] > 0 and rf["algorithmic_bytes_per_launch"] > rf["compulsory_bytes"]


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
