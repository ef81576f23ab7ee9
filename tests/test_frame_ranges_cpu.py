"""_pack.fill_ranges (round 6): the (To, Similarity) columns of a big match filled by helper threads that wait for the result's row
ranges themselves and touch no reference count -- they store similarities and pointers and flag the task; the calling thread follows
the flags and takes one reference per stored pointer.  No device here: `_pack.test_wait` stands in for pfz_event_wait (an int32 flag per range), `_pack.test_set_flags` raises the
flags from a thread of its own while the caller sits inside fill_ranges with the GIL.

Held against the single-threaded fill_columns cell by cell, and against the reference counts a frame must hold: exactly one per
stored pointer, none left when the frame is gone (reference polyfuzz/models/_utils.py:104-125 builds these columns with numpy)."""
import sys

import numpy as np
import pytest

from polyfuzz_amd import _lib
from polyfuzz_amd.models import _utils

pytestmark = pytest.mark.skipif(_lib._pack is None or not hasattr(_lib._pack, "fill_ranges"), reason="_pack.so not built")


def _case(n, top_n, n_names, seed):
    rng = np.random.default_rng(seed)
    names = [f"name {i} é" for i in range(n_names)]
    idx = rng.integers(-1, n_names + 2, (n, top_n)).astype(np.int32)
    near = np.clip(np.arange(n)[:, None] * n_names // max(n, 1) + rng.integers(-3, 4, (n, top_n)), 0, n_names - 1)
    idx = np.where(rng.random((n, top_n)) < 0.7, near, idx).astype(np.int32)         # mostly neighbours, like a sorted list's matches
    val = rng.random((n, top_n)).astype(np.float32)
    val[rng.random((n, top_n)) < 0.2] *= np.float32(0.0015)                          # around the 0.001 / 0.0005 boundaries
    return names, idx, val


def _ends(n, k):
    cuts = sorted({max(1, n * (i + 1) // k) for i in range(k)})
    cuts[-1] = n
    return cuts


def _fill(fb, idx, val, ends, flags, threads, monkeypatch, stamps=None):
    monkeypatch.setattr(_utils, "_RANGE_THREADS", threads)
    fb.fill_ranges(idx.ctypes.data, val.ctypes.data, ends, _lib._pack.test_wait_addr(), flags.ctypes.data, 0, stamps)


@pytest.mark.parametrize("threads", [1, 2, 4, 7])
@pytest.mark.parametrize("n,top_n,k", [(1, 1, 1), (5000, 3, 4), (40000, 5, 12), (30001, 2, 5)])
def test_ranges_equal_the_single_threaded_fill(n, top_n, k, threads, monkeypatch):
    names, idx, val = _case(n, top_n, 6000, n + top_n)
    from_list = [f"q{i}" for i in range(n)]
    want = _utils.FrameBuilder(from_list, names, top_n)
    want.fill(idx, val, 0)
    rc0 = [sys.getrefcount(s) for s in names]
    ends = _ends(n, k)
    flags = np.ones(len(ends), np.int32)
    fb = _utils.FrameBuilder(from_list, names, top_n)
    _fill(fb, idx, val, ends, flags, threads, monkeypatch)
    rc1 = [sys.getrefcount(s) for s in names]
    keep = (np.round(val.astype(np.float64), 3) >= 0.001) & (idx >= 0) & (idx < len(names))
    stored = np.bincount(idx[keep].ravel(), minlength=len(names))
    assert [b - a for a, b in zip(rc0, rc1)] == stored.tolist()               # one reference per stored pointer, exactly
    for r in range(top_n):
        np.testing.assert_array_equal(fb.sims[r], want.sims[r])
        assert all(a is b for a, b in zip(fb.names[r], want.names[r]))
    frame = fb.frame()
    assert frame["To"].tolist() == want.frame()["To"].tolist()
    del frame, fb
    rc2 = [sys.getrefcount(s) for s in names]
    assert rc2 == rc0


def test_ranges_that_arrive_late_and_one_object_in_every_slot(monkeypatch):
    """the flags are raised 300 us apart from another thread while four threads sit in fill_ranges; every cell points at ONE name"""
    n, top_n = 24000, 5
    names = ["the one"] + [f"other {i}" for i in range(99)]
    idx = np.zeros((n, top_n), np.int32)
    val = np.full((n, top_n), 0.5, np.float32)
    ends = _ends(n, 12)
    flags = np.zeros(len(ends), np.int32)
    stamps = np.zeros(2 * len(ends))
    none0 = sys.getrefcount(None)
    fb = _utils.FrameBuilder([f"q{i}" for i in range(n)], names, top_n)
    before = sys.getrefcount(names[0])
    _lib._pack.test_set_flags(flags.ctypes.data, len(ends), 300, 1)
    _fill(fb, idx, val, ends, flags, 4, monkeypatch, stamps)
    during = sys.getrefcount(names[0])
    assert during - before == n * top_n
    assert all(fb.names[r][i] is names[0] for r in range(top_n) for i in (0, 1, n // 2, n - 1))
    seen, filled = stamps[0::2], stamps[1::2]
    # the ranges were waited for (a stamp is taken by whichever thread gets to it first: on a busy box not in range order)
    assert np.all(seen > 0) and np.all(filled >= seen) and seen.max() - seen.min() > 11 * 250e-6
    del fb
    after = sys.getrefcount(names[0])            # (outside the assert: pytest's rewriting keeps the operand alive)
    assert after == before
    assert abs(sys.getrefcount(None) - none0) < 50


@pytest.mark.parametrize("threads", [1, 4])
def test_the_from_column_and_one_object_at_many_list_positions(threads, monkeypatch):
    """a self-match's frame: the From column is filled in the same call (from_pending), and the list holds ONE object at fifty
    positions"""
    n, top_n = 30000, 4
    dup = "a name that is there fifty times"
    names = [f"name {i}" for i in range(n)]
    for i in range(0, n, n // 50):
        names[i] = dup
    rng = np.random.default_rng(3)
    idx = np.clip(np.arange(n)[:, None] + rng.integers(-40, 41, (n, top_n)), 0, n - 1).astype(np.int32)
    val = rng.random((n, top_n)).astype(np.float32)
    rc0 = [sys.getrefcount(s) for s in names]
    dup0 = sys.getrefcount(dup)
    ends = _ends(n, 6)
    fb = _utils.FrameBuilder(names, names, top_n, from_pending=True)
    assert fb.from_pending and all(o is None for o in fb.from_col[:5])
    _fill(fb, idx, val, ends, np.ones(len(ends), np.int32), threads, monkeypatch)
    assert not fb.from_pending and all(a is b for a, b in zip(fb.from_col, names))
    keep = np.round(val.astype(np.float64), 3) >= 0.001
    stored = np.bincount(idx[keep].ravel(), minlength=n) + 1                   # (+ 1: the From column)
    per_object = {}
    for i, s_ in enumerate(names):
        per_object[id(s_)] = per_object.get(id(s_), 0) + int(stored[i])
    want = [per_object[id(s_)] for s_ in names]
    del s_
    rc1 = [sys.getrefcount(s) for s in names]
    assert [b - a for a, b in zip(rc0, rc1)] == want
    frame = fb.frame()
    assert frame["From"].tolist() == names and frame["To_4"].tolist() == [names[j] if k else None for j, k in zip(idx[:, 3], keep[:, 3])]
    del frame, fb
    dup1 = sys.getrefcount(dup)
    assert [sys.getrefcount(s) for s in names] == rc0 and dup1 == dup0
    # a builder whose From column nobody filled fills it when the frame is asked for
    fb = _utils.FrameBuilder(names[:9000], names, 1, from_pending=True)
    fb.fill(idx[:9000, :1], val[:9000, :1], 0)
    assert fb.frame()["From"].tolist() == names[:9000]


@pytest.mark.parametrize("k,fails_at", [(4, 2), (8, 1), (8, 6), (8, 7)])
def test_a_failing_wait_leaves_consistent_columns(k, fails_at, monkeypatch):
    """one of the ranges never arrives (the wait returns an error) -- an early one, a late one, the last one --:
    RuntimeError, and what had been stored by then holds exactly the references it should: dropping the columns brings every
    count back"""
    n, top_n = 20000, 3
    names, idx, val = _case(n, top_n, 3000, 5)
    ends = _ends(n, k)
    flags = np.ones(k, np.int32)
    flags[fails_at] = -1
    rc0 = [sys.getrefcount(s) for s in names]
    fb = _utils.FrameBuilder([f"q{i}" for i in range(n)], names, top_n)
    with pytest.raises(RuntimeError):
        _fill(fb, idx, val, ends, flags, 3, monkeypatch)
    held = np.zeros(len(names), np.int64)
    where = {id(s): i for i, s in enumerate(names)}
    for col in fb.names:
        for o in col:
            if o is not None:
                held[where[id(o)]] += 1
        assert all(o is None for o in col[ends[fails_at - 1]:ends[fails_at - 1] + 5])           # (nothing of the range that failed)
    rc1 = [sys.getrefcount(s) for s in names]
    assert (np.array(rc1) - np.array(rc0) == held).all()                  # exactly one reference per stored pointer
    del fb, col, o
    assert [sys.getrefcount(s) for s in names] == rc0


def test_columns_that_are_not_fresh_are_refused(monkeypatch):
    names, idx, val = _case(3000, 2, 100, 9)
    fb = _utils.FrameBuilder([f"q{i}" for i in range(3000)], names, 2)
    fb.names[1][2999] = "in use"
    with pytest.raises(ValueError):
        _fill(fb, idx, val, [3000], np.ones(1, np.int32), 2, monkeypatch)
    with pytest.raises(ValueError):
        _lib._pack.fill_ranges(names, idx.ctypes.data, val.ctypes.data, 2, (1, 2), (3, 4), (10, 5), _lib._pack.test_wait_addr(), 0, 0, 1)


@pytest.mark.parametrize("threads", [2, 4, 7])
def test_pack_into_on_threads_equals_the_single_walk(threads):
    """pack_into with n_threads > 1 (no From column: that is fill_ranges' business then): two walks per thread over its stretch of
    the list, the stretches' totals added up between them -- same offsets, same characters as the single walk; a wide string, a
    non-str item anywhere in the list, or a buffer that is too small: None"""
    names = [f"name {i} inc é" + "x" * (i % 37) for i in range(40000)] + ["", "y"]
    n = len(names)
    off_bytes = (8 * (n + 1) + 255) & ~255
    one = np.zeros(off_bytes + 48 * n, np.uint8)
    many = np.zeros_like(one)
    rc0 = sys.getrefcount(names[5])
    a = _lib._pack.pack_into(names, 0, one.ctypes.data, off_bytes, len(one))
    b = _lib._pack.pack_into(names, 0, many.ctypes.data, off_bytes, len(many), threads)
    rc1 = sys.getrefcount(names[5])
    assert a == b == sum(len(s) for s in names) and rc1 == rc0
    assert np.array_equal(one[:8 * (n + 1)], many[:8 * (n + 1)]) and np.array_equal(one[off_bytes:off_bytes + a], many[off_bytes:off_bytes + a])
    for bad, cap in ((names[:39000] + ["日本"] + names[39000:], len(many)), (names[:100] + [5] + names[100:], len(many)),
                     (names, off_bytes + a - 1)):
        assert _lib._pack.pack_into(bad, 0, many.ctypes.data, (8 * (len(bad) + 1) + 255) & ~255, cap, threads) is None
    assert _lib._pack.pack_into(names, 0, many.ctypes.data, off_bytes, off_bytes + a, threads) == a          # (fits exactly)


def test_a_whole_frame_of_a_list_against_itself_goes_through_the_crew(monkeypatch):
    """topn_to_frame(idx, val, names, names, top_n) on a big list -- what pipeline.sharded_self_match builds on every rank -- fills
    the To columns AND the From column through fill_ranges (nothing to wait for); same frame as the numpy twin, counts balanced"""
    monkeypatch.setattr(_utils, "_RANGE_THREADS", 4)
    n, top_n = 30000, 3
    names = [f"name {i}" for i in range(n)]
    rng = np.random.default_rng(8)
    idx = np.clip(np.arange(n)[:, None] + rng.integers(-9, 10, (n, top_n)), -1, n).astype(np.int32)
    val = rng.random((n, top_n)).astype(np.float32)
    rc0 = [sys.getrefcount(s) for s in names[:50]]
    a = _utils.topn_to_frame(idx, val, names, names, top_n)
    b = _utils._topn_to_frame_numpy(idx, val, names, names, top_n)
    assert list(a.columns) == list(b.columns) and a["From"].tolist() == names
    for c in a.columns:
        assert (np.array_equal(a[c].to_numpy(), b[c].to_numpy()) if c.startswith("Similarity") else a[c].tolist() == b[c].tolist()), c
    del a, b
    rc1 = [sys.getrefcount(s) for s in names[:50]]
    assert rc1 == rc0


def test_a_forked_child_starts_a_pool_of_its_own():
    """the crews' threads are a pool that sleeps between calls; none of them lives in a forked child -- it must start its own (and
    not wait for the parent's): pack on threads in the parent, fork, pack + fill on threads in the child"""
    import os
    names = [f"name {i} inc" for i in range(40000)]
    n = len(names)
    off_bytes = (8 * (n + 1) + 255) & ~255
    buf = np.zeros(off_bytes + 48 * n, np.uint8)
    want = _lib._pack.pack_into(names, 0, buf.ctypes.data, off_bytes, len(buf), 4)
    assert want == sum(len(s) for s in names)
    pid = os.fork()
    if pid == 0:
        ok = 1
        try:
            buf2 = np.zeros_like(buf)
            got = _lib._pack.pack_into(names, 0, buf2.ctypes.data, off_bytes, len(buf2), 4)
            idx = np.zeros((n, 2), np.int32)
            val = np.full((n, 2), 0.5, np.float32)
            _utils._RANGE_THREADS = 4
            fb = _utils.FrameBuilder(names, names, 2, from_pending=True)
            fb.fill_ranges(idx.ctypes.data, val.ctypes.data, [n], 0, 0, 0)
            ok = 0 if got == want and np.array_equal(buf, buf2) and fb.names[1][n - 1] is names[0] and fb.from_col[5] is names[5] else 2
        finally:
            os._exit(ok)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
