"""every tuning / diagnostic knob of the shipped library that no other test sets is set here once, on a
workload that reaches its code, and must leave the results unchanged (the knobs that are read once per process --
PFZ_SCAN3, PFZ_DEBUG_SYNC -- in a process of their own).  The knobs other tests cover: PFZ_K3_* (test_k3_cossim_gpu.py),
PFZ_K4_PARTS / NO_QUAD / FORCE_GENERAL (test_indel_gpu.py), PFZ_K5_* (test_dense_gpu.py), PFZ_K7_HAND* / PARTS / FORCE_GENERAL
(test_fuzz_gpu.py), PFZ_NO_LDS_HIST, PFZ_K1_EXTRACT (test_vectorize_gpu.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tfidf(names):
    from polyfuzz_amd.models import TFIDF
    return TFIDF(min_similarity=0, top_n=3).match_device(names).download()


def _titles(n_from, n_to):
    from polyfuzz_amd import datasets
    fl, tl = datasets.c3_lists()
    return fl[:n_from], tl[:n_to]


@pytest.mark.parametrize("knob,value", [("PFZ_K1_WAVE_STRINGS", "8"), ("PFZ_K1_WAVE_STRINGS", "64")])
def test_vectoriser_knobs(ctx, monkeypatch, knob, value):
    from polyfuzz_amd import datasets
    names = datasets.load_company_names()[:3000]
    ref = _tfidf(names)
    monkeypatch.setenv(knob, value)
    got = _tfidf(names)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])


def test_lockstep_threshold_knob(ctx, monkeypatch):
    """PFZ_K3_LS_MIN_TO moves the to-side size from which the lock-step kernel serves a match (default 250 000 rows): 20 000 x
    20 000 names with the bar at 1 000 take it -- same result as the row-major kernel."""
    from polyfuzz_amd import datasets, _lib
    names = datasets.load_company_names()
    fl, tl = names[:20000], names[30000:50000]
    f, t = _lib.DeviceStrings.upload(ctx, fl), _lib.DeviceStrings.upload(ctx, tl)
    vec = _lib.DeviceTfidf.fit(ctx, _lib.TfidfParams(3, 3, 1, 1), t, f)
    a, b = vec.transform(f), vec.transform(t)
    ix = _lib.DeviceIndex.build(ctx, b)
    ref = _lib.cossim_topn(ctx, ix, a, 4, 0.0).download()
    monkeypatch.setenv("PFZ_K3_LS_MIN_TO", "1000")
    got = _lib.cossim_topn(ctx, ix, a, 4, 0.0).download()
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])


@pytest.mark.parametrize("knob", ["PFZ_K4_NO_OCTO", "PFZ_K4_SIDE_STREAM"])
def test_edit_distance_knobs(ctx, monkeypatch, knob):
    from polyfuzz_amd import _lib
    fl, tl = _titles(3000, 4000)
    fl = list(fl) + ["x" * 40, "y" * 100]          # (strings of the longer classes: what the side stream is for)
    f, t = _lib.DeviceStrings.upload(ctx, fl), _lib.DeviceStrings.upload(ctx, tl)
    ref = _lib.indel_argmax(ctx, f, t)
    monkeypatch.setenv(knob, "1")
    f2, t2 = _lib.DeviceStrings.upload(ctx, fl), _lib.DeviceStrings.upload(ctx, tl)      # (fresh handles: no cached plan)
    got = _lib.indel_argmax(ctx, f2, t2)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])


def test_dense_panel_bytes_knob(ctx, monkeypatch):
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(4)
    a = rng.standard_normal((700, 96), dtype=np.float32)
    b = rng.standard_normal((5000, 96), dtype=np.float32)
    ref = _lib.dense_cossim_topn_host(ctx, a, b, 5, 0.0, False, normalize=True)
    monkeypatch.setenv("PFZ_K5_PANEL_ROWS", "128")                  # 128-row panels: six of them
    got = _lib.dense_cossim_topn_host(ctx, a, b, 5, 0.0, False, normalize=True)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])


@pytest.mark.parametrize("knob,value", [("PFZ_K7_NO_HANDOVER", "1"), ("PFZ_K7_NO_SIDE_STREAM", "1"),
                                        ("PFZ_K7_ROW_STATS", "/tmp/pfz_k7_rowstats_test.bin")])
def test_fuzz_knobs(ctx, monkeypatch, knob, value):
    """K7's schedule knobs and diagnostics: a list with one very long from-string (the side stream's class) and short heavy ones."""
    from polyfuzz_amd import _lib
    fl, tl = _titles(1500, 6000)
    fl = list(fl) + ["the " * 40 + "end", "a"]
    ref = _lib.fuzz_extract_one(ctx, fl, tl, "WRatio")
    monkeypatch.setenv(knob, value)
    got = _lib.fuzz_extract_one(ctx, fl, tl, "WRatio")
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])


@pytest.mark.parametrize("knob", ["PFZ_SCAN3", "PFZ_DEBUG_SYNC"])
def test_knobs_read_once_per_process(knob):
    """the three-kernel scan (PFZ_SCAN3) and the synchronise-after-every-profiled-kernel mode (PFZ_DEBUG_SYNC) are latched at
    first use: a fresh process with the knob set gives the frame of one without"""
    code = ("import json, sys; sys.path.insert(0, %r)\n"
            "from polyfuzz_amd import datasets\nfrom polyfuzz_amd.models import TFIDF\n"
            "names = datasets.load_company_names()[:6000]\n"
            "df = TFIDF(min_similarity=0, top_n=2).match(names[:3000], names[3000:])\n"
            "print(json.dumps([df['To'].tolist(), df['Similarity'].tolist(), df['To_2'].tolist()]))\n") % REPO
    outs = []
    for env in ({}, {knob: "1"}):
        e = dict(os.environ)
        e.pop(knob, None)
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=e)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1]
