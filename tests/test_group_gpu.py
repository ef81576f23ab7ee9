"""K6 on the GPU: precision_recall_curve against the reference's own outputs (tests/golden/group_golden.json) and the
device single-linkage of a self-match top-1 result against the exact greedy walk (polyfuzz_amd.linkage.single_linkage,
itself pinned on the reference's outputs in tests/test_group_cpu.py) -- all three dicts including their order."""
import json
import os

import numpy as np
import pandas as pd
import pytest

from tests.test_group_cpu import _frame, self_frame

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
AP_TOL = 1e-12      # mean of float64 similarities: the device sums in 2^-44 fixed point, numpy pairwise in float64


@pytest.fixture(scope="module")
def gg():
    with open(os.path.join(HERE, "golden", "group_golden.json")) as f:
        return json.load(f)


def test_precision_recall_curve_vs_reference(gg, ctx):
    from polyfuzz_amd.metrics import precision_recall_curve
    frames = [(_frame(gg[k]["frame"]), gg[k]["pr"]) for k in ("readme", "readme_self")]
    frames.append((self_frame(gg)[0], gg["self"]["pr"]))
    frames += [(_frame(c["frame"]), c["pr"]) for c in gg["random"] + gg["scales"]]
    n = 0
    for df, recs in frames:
        for rec in recs:
            p, r, ap = precision_recall_curve(df, rec["precision_steps"])
            assert p == rec["min_precisions"] and r == rec["recall"]           # thresholds and counts: exact
            exp = np.array([np.nan if x is None else x for x in rec["average_precision"]])
            got = np.array(ap)
            assert np.array_equal(np.isnan(got), np.isnan(exp))
            np.testing.assert_allclose(got[~np.isnan(exp)], exp[~np.isnan(exp)], rtol=0, atol=AP_TOL * max(1.0, df.Similarity.max()))
            n += 1
    assert n >= 20
    with pytest.raises(ZeroDivisionError):
        precision_recall_curve(pd.DataFrame({"From": [], "To": [], "Similarity": []}))


def _same_dicts(a, b):
    for x, y in zip(a, b):
        assert list(x.items()) == list(y.items())


def test_device_linkage_on_arbitrary_top1_graphs(ctx):
    """Rows need not be mutual best matches: random functional graphs, long forward chains (many fixpoint
    rounds), rows without a match -- cluster ids and dict order equal the sequential walk's."""
    from polyfuzz_amd import _lib
    from polyfuzz_amd.linkage import greedy_assign
    rng = np.random.default_rng(3)
    for trial in range(40):
        n = int(rng.integers(2, 400))
        g = rng.integers(0, n, n)
        g = np.where(g == np.arange(n), (g + 1) % n, g)
        if trial % 4 == 1:
            g = np.minimum(np.arange(n) + 1, n - 1)
            g[n - 1] = n - 2                                     # i -> i+1: the recurrence needs ~n/2 rounds
        val = np.round(rng.random(n), 3).astype(np.float32)
        g = np.where(rng.random(n) < 0.1, -1, g)
        val = np.where(g < 0, 0.0, val).astype(np.float32)
        thr = float(rng.choice([0.0, 0.3, 0.75]))
        res = _lib.DeviceTopN.from_host(ctx, g, val)
        cluster, key, info = _lib.linkage_top1(ctx, res, thr)
        kept = (g >= 0) & (np.round(val.astype(np.float64), 3) >= 0.001) & (np.round(val.astype(np.float64), 3) > thr)
        rows = np.nonzero(kept)[0]
        e_cluster, e_order = greedy_assign(rows, g[rows], n)
        np.testing.assert_array_equal(cluster, e_cluster)
        mapped = np.nonzero(cluster >= 0)[0]
        np.testing.assert_array_equal(mapped[np.argsort(key[mapped], kind="stable")], e_order)
        assert (key[cluster < 0] == -1).all()


@pytest.mark.parametrize("thr", [0.0, 0.5, 0.75, 0.9])
def test_group_top1_equals_frame_path(gg, thr):
    from polyfuzz_amd import datasets
    from polyfuzz_amd.linkage import group_top1, single_linkage
    from polyfuzz_amd.models import TFIDF
    lists = [["apple", "apples", "appl", "recal", "house", "similarity"], self_frame(gg)[1],
             list(dict.fromkeys(datasets.load_company_names()[:30000]))]
    for strings in lists:
        strings = list(dict.fromkeys(strings))
        m = TFIDF(n_gram_range=(3, 3), min_similarity=thr)
        _same_dicts(group_top1(m.match_device(strings), strings, thr), single_linkage(m.match(strings), thr))


def test_create_groups_reference_known_answers(gg):
    """reference tests/test_polyfuzz.py:74-100: group() after a two-list match and after a same-list match"""
    from polyfuzz_amd.linkage import create_groups
    from polyfuzz_amd.models import TFIDF
    fl = ["apple", "apples", "appl", "recal", "house", "similarity"]
    tl = ["apple", "apples", "mouse"]
    df = TFIDF(min_similarity=0).match(fl, tl)
    out, clusters, mapping = create_groups(df, link_min_similarity=0.75)
    assert list(out.columns) == ["From", "To", "Similarity", "Group"] and len(out) == 6
    assert clusters == {1: ["apples", "apple"]} and mapping == {"apples": 1, "apple": 1}
    assert out["Group"].tolist()[:3] == ["apples", "apples", "apples"]
    df = TFIDF(min_similarity=0).match(fl, fl)
    out, clusters, mapping = create_groups(df, link_min_similarity=0.75, group_all_strings=True)
    assert clusters == {1: ["apples", "apple", "appl"]} and mapping == {"apples": 1, "apple": 1, "appl": 1}
    # the self-match frame of the 2 000 names, device path vs the reference's dicts for that frame's reference twin
    frame, sl = self_frame(gg)
    strings = list(dict.fromkeys(sl))
    if len(strings) == len(sl):
        for rec in gg["self"]["linkage"]:
            m = TFIDF(n_gram_range=(3, 3), min_similarity=0)
            got = create_groups(pd.DataFrame({"From": sl, "To": sl, "Similarity": 1.0}), m, rec["min_similarity"], True)
            # (near-tie rows may pick another To than the reference's float64 run: compare sizes, not members)
            assert abs(len(got[2]) - len(rec["cluster_mapping"])) <= 0.01 * len(rec["cluster_mapping"]) + 2


def test_k6_edge_cases(ctx):
    """Empty / single-row results, nothing above the threshold, NaN similarities, a non-TFIDF grouper."""
    from polyfuzz_amd import _lib
    from polyfuzz_amd.linkage import create_groups, greedy_assign
    from polyfuzz_amd.metrics import precision_recall_curve
    from polyfuzz_amd.models import EditDistance
    for g, v in (([-1], [0.0]), ([1, 0], [0.2, 0.2]), ([1, 0, 1], [0.9, 0.9, 0.1])):
        res = _lib.DeviceTopN.from_host(ctx, np.array(g, np.int32), np.array(v, np.float32))
        cluster, key, info = _lib.linkage_top1(ctx, res, 0.5)
        rows = [i for i in range(len(g)) if g[i] >= 0 and round(float(np.float32(v[i])), 3) > 0.5]
        e_cluster, _ = greedy_assign(np.array(rows, np.int32), np.array([g[i] for i in rows], np.int32), len(g))
        np.testing.assert_array_equal(cluster, e_cluster)
    with pytest.raises(_lib.PfzError):
        _lib.linkage_top1(ctx, res, -0.1)                    # negative thresholds take the host walk (None can be a key)
    p, r, ap = precision_recall_curve(pd.DataFrame({"From": list("abc"), "To": list("xyz"), "Similarity": [0.5, np.nan, 1.0]}))
    assert r[0] == 2 / 3 and r[-1] == 1 / 3 and abs(ap[0] - 0.75) < 1e-12 and ap[-1] == 1.0      # NaN is never >= a threshold
    df = pd.DataFrame({"From": ["a", "b", "c"], "To": ["apple", "apples", "mouse"], "Similarity": [1.0, 1.0, 1.0]})
    out, clusters, mapping = create_groups(df, EditDistance(normalize=False), link_min_similarity=0.75)
    assert list(out.columns) == ["From", "To", "Similarity", "Group"]                 # scores 0..100: everything links
    assert set(mapping) == {"apple", "apples", "mouse"}
