"""single_linkage against the outputs of the reference's own function (tests/golden/group_golden.json, made by
tests/golden/make_golden_group.py from /root/reference/polyfuzz/linkage.py): the three dicts, INCLUDING their
insertion order, for the reference's test frame at its eleven thresholds, a 2 000-name self-match top-1 frame
(what PolyFuzz._create_groups feeds it) and seeded frames with repeated From strings / None / From == To."""
import json
import os

import numpy as np
import pandas as pd
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gg():
    with open(os.path.join(HERE, "golden", "group_golden.json")) as f:
        return json.load(f)


def self_frame(gg):
    with open(os.path.join(HERE, "golden", "company_self_list.json")) as f:
        sl = json.load(f)["from_list"]
    fr = gg["self"]["frame"]
    to = [None if j < 0 else sl[j] for j in fr["to_index"]]
    return pd.DataFrame({"From": sl, "To": to, "Similarity": fr["Similarity"]}), sl


def _frame(rec):
    return pd.DataFrame({"From": rec["From"], "To": pd.Series(rec["To"], dtype=object), "Similarity": rec["Similarity"]})


def _check(out, exp):
    clusters, mapping, names = out
    assert [[k, v] for k, v in clusters.items()] == exp["clusters"]
    assert [[k, v] for k, v in mapping.items()] == exp["cluster_mapping"]
    assert [[k, v] for k, v in names.items()] == exp["cluster_name_map"]


@pytest.mark.parametrize("helper", ["c", "python"])
def test_single_linkage_equals_reference(gg, helper, monkeypatch):
    from polyfuzz_amd import _lib, linkage
    if helper == "python":
        monkeypatch.setattr(_lib, "_pack", None)
    elif _lib._pack is None:
        pytest.skip("_pack.so not built")
    n = 0
    for name in ("readme", "readme_self"):
        df = _frame(gg[name]["frame"])
        for exp in gg[name]["linkage"]:
            _check(linkage.single_linkage(df, exp["min_similarity"]), exp)
            n += 1
    df, _ = self_frame(gg)
    for exp in gg["self"]["linkage"]:
        _check(linkage.single_linkage(df, exp["min_similarity"]), exp)
        n += 1
    for case in gg["random"]:
        df = _frame(case["frame"])
        for exp in case["linkage"]:
            _check(linkage.single_linkage(df, exp["min_similarity"]), exp)
            n += 1
    assert n >= 60


def test_reference_known_answers(gg):
    """reference tests/test_polyfuzz.py:85-86,99-100 and tests/test_linkage.py:20-31"""
    from polyfuzz_amd.linkage import single_linkage
    df = _frame(gg["readme_self"]["frame"])                  # test_grouper_same_list (test_polyfuzz.py:89-100)
    clusters, mapping, names = single_linkage(df, 0.75)
    assert clusters == {1: ["apples", "apple", "appl"]} and mapping == {"apples": 1, "apple": 1, "appl": 1}
    # test_grouper (test_polyfuzz.py:74-86): the unique To strings apple / apples / mouse matched against themselves;
    # ids start at 1 because cluster 0 is falsy and its two members are re-assigned by the second row
    df = pd.DataFrame({"From": ["apple", "apples", "mouse"], "To": ["apples", "apple", None], "Similarity": [0.8, 0.8, 0.0]})
    clusters, mapping, names = single_linkage(df, 0.75)
    assert clusters == {1: ["apples", "apple"]} and mapping == {"apples": 1, "apple": 1}
    assert names == {"apples": "apples", "apple": "apples"}
    df = _frame(gg["readme"]["frame"])
    assert single_linkage(df, 1.0) == ({}, {}, {})
    for thr in (0.8, 0.9):
        _, mapping, names = single_linkage(df, thr)
        assert max(mapping.values()) == 1 and len(names) == 2
    for thr in (0.6, 0.7):
        _, mapping, names = single_linkage(df, thr)
        assert max(mapping.values()) > 1 and len(names) == 3
