"""Host-side logic that needs no GPU: string packing, DataFrame contract, shard planning, synthetic data."""
import numpy as np
import pandas as pd
import pytest


def test_pack_strings_widths():
    from polyfuzz_amd._lib import pack_strings
    chars, off, w = pack_strings(["ab", "", "cdé"])
    assert w == 1 and chars.dtype == np.uint8 and off.tolist() == [0, 2, 2, 5] and chars[-1] == 0xE9
    chars, off, w = pack_strings(["ab", "한글", ""])
    assert w == 4 and chars.dtype == np.uint32 and off.tolist() == [0, 2, 4, 4] and chars[2] == ord("한")
    chars, off, w = pack_strings([])
    assert len(chars) == 0 and off.tolist() == [0]


def test_c_packer_equals_python_packer():
    """polyfuzz_amd/_pack.so (CPython helper) and its pure-Python twin give identical buffers."""
    from polyfuzz_amd import _lib
    if _lib._pack is None:
        pytest.skip("_pack.so not built")
    rng = np.random.default_rng(4)
    pools = ["abc XYZ-09 ", "a\u00f1\u00e9\u00fc ", "\u65e5\u672c\u8a9e ab", "x\U0001f600y", "a\ud800b"]   # latin-1, wide, astral, lone surrogate
    for pool in pools:
        chars = np.array(list(pool), dtype=object)
        strings = ["".join(rng.choice(chars, size=int(rng.integers(0, 12))).tolist()) for _ in range(300)] + ["", pool]
        a = _lib.pack_strings(strings)
        b = _lib._pack_strings_py(strings)
        assert a[2] == b[2]
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    assert _lib.pack_strings([])[1].tolist() == [0]
    assert _lib.pack_strings(("ab", "c"))[1].tolist() == [0, 2, 3]        # tuples too
    with pytest.raises(TypeError):
        _lib.pack_strings(["ok", 3])


def test_topn_to_frame_contract():
    """reference _utils.py:104-125: column order, 3-dp rounding, < 0.001 -> 0.0 / None, -1 -> None."""
    from polyfuzz_amd.models._utils import topn_to_frame
    idx = np.array([[1, 0], [0, -1], [-1, -1]], np.int32)
    val = np.array([[0.78375, 0.0004], [1.0, 0.0], [0.0, 0.0]], np.float32)
    df = topn_to_frame(idx, val, ["a", "b", "c"], ["x", "y"], 2)
    assert list(df.columns) == ["From", "To", "Similarity", "To_2", "Similarity_2"]
    assert df["To"].tolist() == ["y", "x", None] and df["To_2"].tolist() == [None, None, None]
    np.testing.assert_allclose(df["Similarity"], [0.784, 1.0, 0.0])
    assert df["Similarity_2"].tolist() == [0.0, 0.0, 0.0]
    assert df["Similarity"].dtype == np.float64 and df["From"].tolist() == ["a", "b", "c"]


def test_clean_string_matches_oracle(oracle_mod):
    from polyfuzz_amd.models._tfidf import _clean_string
    for s in ["  Hello,   World!! ", "A\tB  C\nD", "İstanbul", "Ünited", "", "K2 — the 2nd"]:
        assert _clean_string(s) == oracle_mod.clean_string(s)


def test_edit_distance_scorer_gate():
    from polyfuzz_amd.models import EditDistance, BaseMatcher
    assert isinstance(EditDistance(), BaseMatcher) and EditDistance().type == "EditDistance"
    with pytest.raises(NotImplementedError):
        EditDistance(scorer=len)

    class ratio:       # looks like rapidfuzz.fuzz.ratio
        __name__ = "ratio"
        __module__ = "rapidfuzz.fuzz"
    EditDistance(scorer=ratio())


def test_rapidfuzz_scorer_gate_checks_the_module():
    """a callable merely NAMED like a rapidfuzz scorer (Levenshtein.ratio, a user's own WRatio) has no kernel"""
    from polyfuzz_amd.models import RapidFuzz

    def ratio(a, b):
        return 1.0
    with pytest.raises(NotImplementedError):
        RapidFuzz(scorer=ratio)

    class WRatio:
        __name__ = "WRatio"
        __module__ = "rapidfuzz.fuzz_py"
    assert RapidFuzz(scorer=WRatio())._scorer_name == "WRatio"
    assert RapidFuzz()._scorer_name == "WRatio" and RapidFuzz(scorer="token_set_ratio")._scorer_name == "token_set_ratio"


def test_base_matcher_is_abstract():
    from polyfuzz_amd.models import BaseMatcher, TFIDF
    with pytest.raises(TypeError):
        BaseMatcher()
    m = TFIDF(n_gram_range=(2, 3), min_similarity=0.5, top_n=3, model_id="m")
    assert (m.type, m.model_id, m.top_n, m.cosine_method) == ("TF-IDF", "m", 3, "sparse")


def test_shard_bounds_cover_everything():
    from polyfuzz_amd.pipeline import shard_bounds
    for n in (0, 1, 7, 100, 100003):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(n, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1


def test_clip_top_n_equals_reference_rule():
    from polyfuzz_amd.models._utils import clip_top_n
    for to_list in (["a", "b", "a"], ["x"], [], ["a"] * 5, list("abcdefg")):
        for top_n in (1, 2, 3, 10):
            assert clip_top_n(top_n, to_list) == min(top_n, len(set(to_list)))     # reference _utils.py:54-56
    assert clip_top_n(7, None) == 7


def test_unpad_gathered_shards():
    from polyfuzz_amd.pipeline import TfidfMatchJob, shard_bounds
    n, world, top_n = 11, 3, 2
    sizes = [shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world)]
    rpr = max(sizes)
    idx = np.full((world * rpr, top_n), -1, np.int32)
    val = np.zeros((world * rpr, top_n), np.float32)
    row = 0
    for r, sz in enumerate(sizes):
        for i in range(sz):
            idx[r * rpr + i] = row
            val[r * rpr + i] = row / 10
            row += 1
    i2, v2 = TfidfMatchJob.unpad(idx, val, sizes, rpr)
    assert i2.shape == (n, top_n) and (i2[:, 0] == np.arange(n)).all() and np.allclose(v2[:, 1], np.arange(n) / 10)


def test_synthetic_names_are_deterministic_and_name_like():
    from polyfuzz_amd import synth
    a, b = synth.company_names(500, 7), synth.company_names(500, 7)
    assert a == b and a != synth.company_names(500, 8)
    assert 15 < np.mean([len(s) for s in a]) < 35


@pytest.mark.parametrize("threads", [1, 4])
def test_frame_helper_equals_numpy_twin(threads, monkeypatch):
    """_pack.fill_columns (C, the calling thread) / _pack.fill_ranges (a big fill: the crew of host threads) build exactly the frame of
    the numpy twin -- np.round(., 3), the < 0.001 rule at the rounding boundary, -1 / out-of-range rows --
    and leaves every reference count balanced; a frame filled in row ranges from raw addresses (what TFIDF.match does under the
    device's work, from the context's pinned memory) is the same frame."""
    import sys
    from polyfuzz_amd import _lib
    from polyfuzz_amd.models import _utils
    if _lib._pack is None:
        pytest.skip("_pack.so not built")
    monkeypatch.setattr(_utils, "_RANGE_THREADS", threads)
    rng = np.random.default_rng(11)
    to_list = [f"name {i} é" for i in range(5000)]
    for n, top_n in ((0, 1), (7, 3), (30000, 5)):
        from_list = [f"q{i}" for i in range(n)]
        idx = rng.integers(-1, len(to_list) + 2, (n, top_n)).astype(np.int32)
        val = rng.random((n, top_n)).astype(np.float32)
        val[rng.random((n, top_n)) < 0.2] *= np.float32(0.0015)              # around the 0.001 / 0.0005 boundaries
        if n:
            val[0, 0], val[1 % n, 0] = np.float32(0.0005), np.float32(0.00049999)
        a = _utils.topn_to_frame(idx, val, from_list, to_list, top_n)
        b = _utils._topn_to_frame_numpy(idx, val, from_list, to_list, top_n)
        assert list(a.columns) == list(b.columns) and a.dtypes.tolist() == b.dtypes.tolist()
        for c in a.columns:
            if c.startswith("Similarity"):
                np.testing.assert_array_equal(a[c].to_numpy(), b[c].to_numpy())
            else:
                assert a[c].tolist() == b[c].tolist(), c

    # in row ranges, each from raw addresses (the context's pinned staging in TFIDF.match)
    none0 = sys.getrefcount(None)
    fb = _utils.FrameBuilder(from_list, to_list, top_n)
    for lo, hi in ((0, 9000), (9000, 9001), (9001, n)):
        i2, v2 = np.ascontiguousarray(idx[lo:hi]), np.ascontiguousarray(val[lo:hi])
        fb.fill_raw(i2.ctypes.data, v2.ctypes.data, hi - lo, lo)
    c = fb.frame()
    for col in a.columns:
        assert (np.array_equal(c[col].to_numpy(), a[col].to_numpy()) if col.startswith("Similarity") else c[col].tolist() == a[col].tolist())
    del c, fb
    assert abs(sys.getrefcount(None) - none0) < 50            # (None's count is the interpreter's too: balanced up to its own traffic)

    def held(i):          # references a frame holds on to_list[i] while alive / after it is gone
        before = sys.getrefcount(to_list[i])
        frame = _utils.topn_to_frame(idx, val, from_list, to_list, top_n)
        during = sys.getrefcount(to_list[i])
        del frame
        return during - before, sys.getrefcount(to_list[i]) - before
    during, after = held(17)
    assert after == 0 and during == int(((idx == 17) & (np.round(val.astype(np.float64), 3) >= 0.001)).sum())
    # the few-queries-against-a-long-list path takes the list as it is (no object-array copy of it)
    df = _utils.topn_to_frame(np.array([[4999]], np.int32), np.array([[0.5]], np.float32), ["q"], to_list, 1)
    assert df["To"].tolist() == [to_list[4999]]


def test_balanced_bounds_cover_everything_and_balance_characters():
    """pipeline.balanced_bounds: contiguous, covering, monotone cuts with (nearly) equal character counts; degenerate lists."""
    from polyfuzz_amd.pipeline import balanced_bounds
    rng = np.random.default_rng(3)
    strings = ["x" * int(k) for k in np.sort(rng.integers(1, 90, 5000))]      # sorted = skewed, like the reference's lists
    for world in (1, 2, 3, 8):
        bounds = balanced_bounds(strings, world)
        assert len(bounds) == world and bounds[0][0] == 0 and bounds[-1][1] == len(strings)
        assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:])) and all(b <= e for b, e in bounds)
        chars = [sum(len(s) + 1 for s in strings[b:e]) for b, e in bounds]
        assert max(chars) <= 1.02 * (sum(chars) / world) + 100
    assert balanced_bounds([], 4) == [(0, 0)] * 4
    assert [e - b for b, e in balanced_bounds(["a"], 3)] .count(1) == 1
    assert sum(e - b for b, e in balanced_bounds(["", "", ""], 2)) == 3


def test_frame_beyond_1024_match_columns():
    """top_n has no device-side limit any more (passes of 1024): the CPython frame helper keeps up -- 1 100 (To, Similarity)
    column pairs from the heap instead of its 1024-entry stack arrays, equal to the numpy twin."""
    from polyfuzz_amd import _lib
    from polyfuzz_amd.models._utils import topn_to_frame, _topn_to_frame_numpy
    if _lib._pack is None:
        pytest.skip("_pack.so not built")
    rng = np.random.default_rng(0)
    names = [f"n{i}" for i in range(50)]
    n, top = 40, 1100
    idx = rng.integers(-1, 50, (n, top)).astype(np.int32)
    val = rng.random((n, top)).astype(np.float32)
    val[idx < 0] = 0
    a = topn_to_frame(idx, val, names[:n], names, top)
    b = _topn_to_frame_numpy(idx, val, names[:n], names, top)
    assert a.shape == (n, 1 + 2 * top) and a.equals(b)


@pytest.mark.parametrize("threads", [1, 4, -4])        # (negative: the general two-pass path also for 1-byte lists)
def test_packer_fills_the_from_column_in_its_own_walk(threads, monkeypatch):
    """_pack.pack(strings, n_threads, obj_addr): the frame's From column (a fresh object array) is filled while the strings
    are packed for their upload -- same strings, one new reference each, nothing else changed; wide (UTF-32) lists, lists
    that need a second start (not-ready strings cannot be made here, a non-str item can) and a used array are refused."""
    import sys
    from polyfuzz_amd import _lib
    if _lib._pack is None:
        pytest.skip("_pack.so not built")
    monkeypatch.setattr(_lib, "_PACK_THREADS", threads)
    for names in ([f"name {i} inc" for i in range(40000)] + ["dup"] * 50, ["é", "日本語", "x" * 300] + [f"n{i}" for i in range(20000)]):
        rc0 = [sys.getrefcount(s) for s in names[:5]]
        last = names[-1]
        last0 = sys.getrefcount(last)
        col = np.empty(len(names), dtype=object)
        a = _lib.pack_strings(names, col)
        b = _lib.pack_strings(names)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
        assert all(col[i] is names[i] for i in range(len(names)))
        assert [sys.getrefcount(s) for s in names[:5]] == [r + 1 for r in rc0]
        assert sys.getrefcount(last) - last0 == names.count(last)     # ("dup": one object in 50 slots, atomic adds across the workers)
        del col
        assert [sys.getrefcount(s) for s in names[:5]] == rc0 and sys.getrefcount(last) == last0
    used = np.empty(3, dtype=object)
    used[1] = "x"
    with pytest.raises(ValueError):
        _lib.pack_strings(["a", "b", "c"], used)
    with pytest.raises(TypeError):
        _lib.pack_strings(["a", 5, "c"], np.empty(3, dtype=object))


@pytest.mark.parametrize("n,top_n", [(0, 1), (1, 1), (1, 3), (5, 2), (300, 4)])
def test_small_frames_put_together_from_their_blocks_equal_the_constructor_s(n, top_n, monkeypatch):
    """frames of fewer than 8 192 rows skip pd.DataFrame(dict) (a third of a single query's wall time): one object block, one
    float64 block, a RangeIndex, from pandas' own parts -- checked against the constructor once per process, and here:
    equal frames (values, dtypes, columns, index type), ordinary behaviour afterwards (assignment, a new column, to_dict),
    reference counts of the names balanced; PFZ_FAST_FRAME=0 is the constructor's frame."""
    import sys
    import pandas as pd
    from polyfuzz_amd import _lib
    from polyfuzz_amd.models import _utils
    if _lib._pack is None:
        pytest.skip("_pack.so not built")
    rng = np.random.default_rng(n * 7 + top_n)
    to_list = [f"name {i}" for i in range(50)]
    from_list = [f"q{i}" for i in range(n)]
    idx = rng.integers(-1, len(to_list) + 1, (n, top_n)).astype(np.int32)
    val = rng.random((n, top_n)).astype(np.float32)
    val[rng.random((n, top_n)) < 0.3] = np.float32(0.0004)
    monkeypatch.setitem(_utils._FAST_FRAME, "ok", None)
    assert _utils._fast_frame_ok()
    before = sys.getrefcount(to_list[7])
    fast = _utils.topn_to_frame(idx, val, from_list, to_list, top_n)
    monkeypatch.setenv("PFZ_FAST_FRAME", "0")
    monkeypatch.setitem(_utils._FAST_FRAME, "ok", None)
    slow = _utils.topn_to_frame(idx, val, from_list, to_list, top_n)
    assert not _utils._FAST_FRAME["ok"]
    pd.testing.assert_frame_equal(fast, slow)
    assert isinstance(fast.index, pd.RangeIndex) and fast.dtypes.tolist() == slow.dtypes.tolist()
    assert fast.to_dict() == slow.to_dict()
    if n:
        fast.loc[0, "Similarity"] = 0.125
        fast["extra"] = np.arange(n)
        assert fast["Similarity"].iloc[0] == 0.125 and list(fast.columns)[-1] == "extra" and len(fast.columns) == 2 + 2 * top_n
        assert pd.concat([fast, fast]).shape == (2 * n, 2 + 2 * top_n)
    del fast, slow
    after = sys.getrefcount(to_list[7])          # (outside the assert: pytest's rewriting keeps the operand alive)
    assert after == before


def test_pack_into_a_caller_s_buffer_equals_pack():
    """_pack.pack_into (round 6: the string packer writing straight into the engine's pinned staging buffer): offsets and
    characters equal to pack()'s, the From column filled in the same walk, reference counts exact; a wide string, a non-str item
    or a buffer that is too small give None and leave nothing behind."""
    import sys
    from polyfuzz_amd import _lib
    if _lib._pack is None or not hasattr(_lib._pack, "pack_into"):
        pytest.skip("_pack.so not built")
    names = [f"name {i} inc é" for i in range(5000)] + ["", "x"]
    n = len(names)
    raw, off, width = _lib._pack.pack(names, 1)
    assert width == 1
    off_bytes = (8 * (n + 1) + 255) & ~255
    buf = np.zeros(off_bytes + len(raw) + 64, np.uint8)
    col = np.empty(n, dtype=object)
    rc0 = sys.getrefcount(names[3])
    got = _lib._pack.pack_into(names, col.ctypes.data, buf.ctypes.data, off_bytes, len(buf))
    assert got == len(raw)
    assert buf[:8 * (n + 1)].tobytes() == off and buf[off_bytes:off_bytes + len(raw)].tobytes() == raw
    rc1 = sys.getrefcount(names[3])            # (outside the asserts: pytest's rewriting keeps operands in temporaries)
    assert all(col[i] is names[i] for i in range(n)) and rc1 == rc0 + 1
    del col
    rc2 = sys.getrefcount(names[3])
    assert rc2 == rc0
    for bad, cap in ((names[:10] + ["日本"] + names[10:20], len(buf)), (names[:10] + [5], len(buf)), (names, off_bytes + 100)):
        col = np.empty(len(bad), dtype=object)
        rc0 = sys.getrefcount(bad[3])
        assert _lib._pack.pack_into(bad, col.ctypes.data, buf.ctypes.data, off_bytes, cap) is None
        rc1 = sys.getrefcount(bad[3])
        assert all(c is None for c in col) and rc1 == rc0
    with pytest.raises(ValueError):
        _lib._pack.pack_into(names, 0, buf.ctypes.data, 8, len(buf))


def test_pair_frames_from_their_blocks_equal_the_constructor_s():
    """The From / To / Similarity frame of EditDistance / RapidFuzz put together from two blocks that exist before the device's result
    does (pair_frame_blocks / pair_frame: no pandas constructor, no copy), and a TFIDF frame of fewer rows than a split match has
    living in its blocks from the start (FrameBuilder): equal to the constructor's frames -- values, dtypes, columns, index --,
    ordinary frames afterwards, the names' reference counts balanced."""
    import sys
    import pandas as pd
    from polyfuzz_amd import _lib
    from polyfuzz_amd.models import _utils
    if _lib._pack is None or not _utils._fast_frame_ok():
        pytest.skip("_pack.so not built / this pandas' parts differ")
    rng = np.random.default_rng(5)
    names = [f"name {i}" for i in range(300)]
    from_list = [f"q{i}" for i in range(12000)]
    n = len(from_list)
    idx = rng.integers(-1, len(names), n).astype(np.int32)
    sim = rng.random(n)
    keep = rng.random(n) < 0.8
    rc0 = sys.getrefcount(names[7])
    for k in (None, keep):
        fast = _utils.pair_frame(from_list, names, idx, sim, keep=k, blocks=_utils.pair_frame_blocks(from_list))
        slow = _utils.pair_frame(from_list, names, idx, sim, keep=k)
        pd.testing.assert_frame_equal(fast, slow)
        assert list(fast.columns) == ["From", "To", "Similarity"] and isinstance(fast.index, pd.RangeIndex)
        fast["Similarity"] = (fast["Similarity"] - fast["Similarity"].min()) / (fast["Similarity"].max() - fast["Similarity"].min())
        assert fast["Similarity"].max() == 1.0 and fast["To"].isna().sum() == int(((idx < 0) | (~k if k is not None else False)).sum())
        del fast, slow
    rc1 = sys.getrefcount(names[7])
    assert rc1 == rc0
    assert _utils.pair_frame_blocks(tuple(from_list)) is not None and _utils.pair_frame_blocks(np.array(from_list, dtype=object)) is None
    # a TFIDF frame of 12 000 rows: blocks from the start
    top_n = 3
    i2 = rng.integers(-1, len(names) + 1, (n, top_n)).astype(np.int32)
    v2 = rng.random((n, top_n)).astype(np.float32)
    fb = _utils.FrameBuilder(from_list, names, top_n)
    assert fb._blocks is not None and fb.names[0].base is fb._blocks[0]
    fb.fill(i2, v2, 0)
    a, b = fb.frame(), _utils._topn_to_frame_numpy(i2, v2, from_list, names, top_n)
    pd.testing.assert_frame_equal(a, b)
    assert np.shares_memory(a["To_2"].to_numpy(), fb._blocks[0])
    del a, b, fb
    rc2 = sys.getrefcount(names[7])
    assert rc2 == rc0
