import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build_native()
    return oracle


@pytest.fixture(scope="session")
def ctx():
    """A device context; GPU tests fail (not skip) when the library or device is missing."""
    import polyfuzz_amd
    return polyfuzz_amd.Context.default()


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    g = os.path.join(REPO, "tests", "golden")
    out = {"npz": np.load(os.path.join(g, "golden.npz"))}
    for name in ("readme_cases", "company_c2_lists", "company_self_list", "titles_lists", "titles_self_list"):
        with open(os.path.join(g, name + ".json")) as f:
            out[name] = json.load(f)
    with open(os.path.join(g, "lcs_golden.json"), encoding="utf-8") as f:
        out["lcs_golden"] = json.load(f)          # textdistance.lcsseq + nltk.edit_distance (make_golden_lcs.py)
    return out
