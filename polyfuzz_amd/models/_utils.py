"""cosine_similarity -- the similarity operator of the reference
(polyfuzz/models/_utils.py:15-125) on the MI355X engine.

Same signature and output contract; every back-end name the reference accepts
("sparse", "sklearn", "knn") runs the HIP kernel (K3 for sparse input, K5 for
dense ndarray input) -- they differ only in what the reference makes them differ
in: "sparse" honours `min_similarity` (strict >), "sklearn"/"knn" ignore it
(_utils.py:62-68,95 vs :82).  There is no CPU path.
"""
import os
from typing import List

import numpy as np
import pandas as pd
from scipy.sparse import issparse

from .. import _lib

_METHODS = ("sparse", "sklearn", "knn", "hip")
# host threads of a big frame's gathers (_pack.fill_ranges), the calling thread included: _lib.host_threads() -- PFZ_HOST_THREADS
_RANGE_THREADS = _lib.host_threads()


_BLOCK_FRAME_ROWS = 20000       # frames of fewer rows live in their two blocks from the start (FrameBuilder); = TFIDF's _SPLIT_MIN_ROWS


def object_column(strings, out=None) -> np.ndarray:
    """list[str] -> 1-D object ndarray (the From column; built while the GPU is still busy).  out: a fresh, contiguous object array
    of that length to fill instead"""
    arr = np.empty(len(strings), dtype=object) if out is None else out
    if _lib._pack is not None and isinstance(strings, (list, tuple)) and hasattr(_lib._pack, "fill_objects"):
        _lib._pack.fill_objects(strings, arr.ctypes.data, len(strings))      # (numpy's own assignment is 10x slower)
    else:
        arr[:] = strings
    return arr


def gather_column(names, idx, keep=None, out=None) -> np.ndarray:
    """object column of names[idx[i]] (None where `keep[i]` is false or idx[i] < 0): one pass of the CPython helper
    (prefetched gathers of the names), or -- without it -- one fancy-index over an object pool.  out: a fresh, contiguous object
    array of len(idx) to fill instead"""
    if _lib._pack is not None and isinstance(names, (list, tuple)) and hasattr(_lib._pack, "gather_objects"):
        j = np.ascontiguousarray(idx, np.int32)
        out = np.empty(len(j), dtype=object) if out is None else out
        k = None if keep is None else np.ascontiguousarray(keep, np.uint8)
        _lib._pack.gather_objects(names, j.ctypes.data, len(j), out.ctypes.data, 0 if k is None else k.ctypes.data)
        return out
    pool = np.empty(len(names) + 1, dtype=object)
    pool[:len(names)] = object_column(names)
    pool[len(names)] = None
    j = np.asarray(idx, np.int64)
    bad = j < 0 if keep is None else (j < 0) | ~np.asarray(keep, bool)
    col = pool[np.where(bad, len(names), j)]
    if out is None:
        return col
    out[:] = col
    return out


def pair_frame(from_list, names, idx, sim, keep=None, from_col=None, blocks=None) -> pd.DataFrame:
    """The From / To / Similarity frame of the best-match matchers (reference _distance.py:77-81, _rapidfuzz.py:84-91): To = names[idx]
    (None where idx < 0 or `keep` is false), Similarity = sim.  blocks: what pair_frame_blocks() returned -- the From column is
    in its place already (blocks[0][0], filled while the device worked) and the frame is put together from the two blocks without
    pandas' constructor and without a copy; None: the constructor's frame."""
    if blocks is None:
        if from_col is None:
            from_col = object_column(from_list)
        return pd.DataFrame({"From": from_col, "To": gather_column(names, idx, keep), "Similarity": sim}, copy=False)
    obj, flt = blocks
    gather_column(names, idx, keep, out=obj[1])
    flt[0] = sim
    return _fast_frame_of_blocks(obj, flt)


def pair_frame_blocks(from_list):
    """the two blocks of a From / To / Similarity frame with the From column filled in -- (object[2][n], float64[1][n]) --, or None
    where the block shortcut does not apply (no CPython helper, a from-list that is no list, a pandas whose parts differ)"""
    if _lib._pack is None or not isinstance(from_list, (list, tuple)) or not _fast_frame_ok():
        return None
    n = len(from_list)
    obj, flt = np.empty((2, n), dtype=object), np.empty((1, n), np.float64)
    object_column(from_list, out=obj[0])
    return obj, flt


# ---- a small result frame without pandas' constructor -----------------------------------------------------------------
# pd.DataFrame(dict) looks at every column (sanitize, infer, consolidate): ~35 us for the 1 x 3 frame of a single query,
# a third of that match's wall time.  The frame below is the same object -- one object block (From, To, To_2 ...), one float64
# block (the similarities), a RangeIndex -- put together from its parts.  The parts are pandas internals, so the shortcut
# proves itself against the constructor once per process (frames equal, same dtypes, same columns) and is left alone if it
# cannot: then every frame comes from the constructor.
_FAST_FRAME = {"ok": None, "parts": {}}


def _fast_frame_parts(top_n):
    parts = _FAST_FRAME["parts"].get(top_n)
    if parts is None:
        from pandas._libs.internals import BlockPlacement
        cols = ["From"]
        for r in range(top_n):
            cols += ["To" if r == 0 else f"To_{r + 1}", "Similarity" if r == 0 else f"Similarity_{r + 1}"]
        obj_at = np.array([0] + [1 + 2 * r for r in range(top_n)], np.intp)
        flt_at = np.array([2 + 2 * r for r in range(top_n)], np.intp)
        parts = (pd.Index(cols), BlockPlacement(obj_at), BlockPlacement(flt_at))
        _FAST_FRAME["parts"][top_n] = parts
    return parts


def _fast_frame(from_col, names, sims):
    from pandas.core.internals.blocks import new_block_2d
    from pandas.core.internals.managers import BlockManager
    n, k = len(from_col), len(names)
    cols, obj_at, flt_at = _fast_frame_parts(k)
    obj = np.empty((1 + k, n), dtype=object)
    obj[0] = from_col
    flt = np.empty((k, n), np.float64)
    for r in range(k):
        obj[1 + r] = names[r]
        flt[r] = sims[r]
    mgr = BlockManager((new_block_2d(obj, obj_at), new_block_2d(flt, flt_at)), [cols, pd.RangeIndex(n)], verify_integrity=False)
    return pd.DataFrame._from_mgr(mgr, axes=mgr.axes)


def _fast_frame_of_blocks(obj, flt):
    """_fast_frame() around blocks that exist already: obj[0] = From, obj[1 + r] = To_r, flt[r] = Similarity_r (no copy)"""
    from pandas.core.internals.blocks import new_block_2d
    from pandas.core.internals.managers import BlockManager
    cols, obj_at, flt_at = _fast_frame_parts(len(flt))
    mgr = BlockManager((new_block_2d(obj, obj_at), new_block_2d(flt, flt_at)), [cols, pd.RangeIndex(obj.shape[1])], verify_integrity=False)
    return pd.DataFrame._from_mgr(mgr, axes=mgr.axes)


def _fast_frame_ok():
    if _FAST_FRAME["ok"] is None:
        ok = False
        if os.environ.get("PFZ_FAST_FRAME", "1") != "0":
            try:
                f = np.array(["a", "b"], dtype=object)
                t = [np.array(["x", None], dtype=object), np.array([None, "y"], dtype=object)]
                v = [np.array([0.5, 0.0]), np.array([0.0, 0.25])]
                fast = _fast_frame(f, t, v)
                slow = pd.DataFrame({"From": f, "To": t[0], "Similarity": v[0], "To_2": t[1], "Similarity_2": v[1]}, copy=False)
                ok = bool(fast.equals(slow) and list(fast.columns) == list(slow.columns) and list(fast.dtypes) == list(slow.dtypes)
                          and fast.index.equals(slow.index) and isinstance(fast.index, pd.RangeIndex))
                if ok:
                    fast.loc[0, "Similarity"] = 1.0          # (an ordinary, writable frame)
                    ok = fast["Similarity"].tolist() == [1.0, 0.0] and fast["To"].tolist() == ["x", None]
            except Exception:
                ok = False
        _FAST_FRAME["ok"] = ok
    return _FAST_FRAME["ok"]


class FrameBuilder:
    """The reference's result frame (_utils.py:104-125) built in row ranges: `fill(idx, val, row0)` writes the To /
    Similarity columns of rows [row0, row0 + len(idx)) -- so the first part of a split match can be turned into
    columns while the device still works on the rest -- and `frame()` wraps the columns."""

    def __init__(self, from_list, to_list, top_n, from_col=None, from_pending=False):
        """from_pending: the From column is left empty here -- fill_ranges' threads fill it from from_list before the first range"""
        self.n, self.top_n, self.to_list = len(from_list), top_n, to_list
        self.from_list = from_list
        self.from_pending = bool(from_pending and from_col is None)
        # A frame that is not built under the device's work (fewer rows than a split match has) lives in its two BLOCKS from the start
        # -- one object array (From, To, To_2 ...) x rows, one float64 array (the similarities) x rows: the columns below are rows of
        # them, the fills write there, and frame() puts the DataFrame together from the blocks (_fast_frame_of_blocks) without
        # pandas' constructor looking at eleven columns (0.3 ms of a 0.75-ms match of 10 000 x 10 000 names) and without a copy.
        self._blocks = None
        if top_n >= 1 and self.n < _BLOCK_FRAME_ROWS and _fast_frame_ok():
            obj = np.empty((1 + top_n, self.n), dtype=object)
            flt = np.empty((top_n, self.n), np.float64)
            self._blocks = (obj, flt)
            if from_col is not None:
                obj[0] = from_col
            elif not self.from_pending:
                object_column(from_list, out=obj[0])
            self.from_col = obj[0]
            self.names = [obj[1 + r] for r in range(top_n)]
            self.sims = [flt[r] for r in range(top_n)]
        else:
            self.from_col = np.empty(self.n, dtype=object) if self.from_pending else object_column(from_list) if from_col is None else from_col
            self.names = [np.empty(self.n, dtype=object) for _ in range(top_n)]
            self.sims = [np.empty(self.n, np.float64) for _ in range(top_n)]
        # (the columns' data addresses, taken ONCE: `ndarray.ctypes` builds a helper object on every access -- ten of them per
        # filled row range were 10 - 18 us of a range's 80)
        self._name_at = [a.ctypes.data for a in self.names]
        self._sim_at = [a.ctypes.data for a in self.sims]
        # The DataFrame is wrapped around the (still empty) columns NOW -- the caller is waiting for the device anyway, and
        # pandas takes ~0.5 ms to look at eleven 100 000-element columns -- and `fill` writes through the arrays it shares
        # with them.  Only if this pandas really shares them (copy=False is a request): otherwise frame() builds it at the end.
        self._frame = None
        if self._blocks is None and self.n >= 8192 and top_n and os.environ.get("PFZ_EARLY_FRAME", "1") != "0":     # (a small frame is made in no time, and the sharing checks below would be most of a single query's host time)
            # (pandas scans an object column for date-likes until it meets a non-null: an all-None column is scanned to its end.
            # The first slot holds a string while the frame is made, and None again before anything is filled in)
            empty = self.names + ([self.from_col] if self.from_pending else [])
            for a in empty:
                a[0] = ""
            f = self._wrap()
            for a in empty:
                a[0] = None
            if np.shares_memory(f["To"].values, self.names[0]) and np.shares_memory(f["Similarity"].values, self.sims[0]) and \
                    np.shares_memory(f.iloc[:, -1].values, self.sims[-1]):
                self._frame = f

    def fill_raw(self, idx_addr, val_addr, m, row0):
        """fill() from the addresses of int32 idx[m][top_n] / fp32 val[m][top_n] (the context's pinned staging: no copy between
        the device's result and the columns).  A big stretch goes to the crew of host threads (_pack.fill_ranges, nothing to wait
        for), a small one is filled by the calling thread (_pack.fill_columns)."""
        if not m or not self.top_n:
            return
        at = 8 * row0
        names_at, sims_at = tuple(b + at for b in self._name_at), tuple(b + at for b in self._sim_at)
        if _RANGE_THREADS > 1 and m * self.top_n >= 65536 and self.top_n <= 1024 and hasattr(_lib._pack, "fill_ranges"):
            _lib._pack.fill_ranges(self.to_list, idx_addr, val_addr, self.top_n, names_at, sims_at, (int(m),), 0, 0, 0, _RANGE_THREADS)
        else:
            _lib._pack.fill_columns(self.to_list, idx_addr, val_addr, m, self.top_n, names_at, sims_at, 1)

    def fill_ranges(self, idx_addr, val_addr, ends, wait_addr, ctx_addr, first_slot, stamps=None):
        """every row range of a result that arrives in ascending ranges [0, ends[0]), [ends[0], ends[1]) ... at idx_addr / val_addr
        (the whole result, pinned host memory): range i is filled once wait(ctx, first_slot + i) has returned (wait_addr 0:
        everything is there) -- _RANGE_THREADS - 1 helper threads wait for the ranges themselves and store similarities and
        pointers, the calling thread takes the references (_pack.fill_ranges).  stamps: float64[2 * len(ends)] or None (seconds on
        time.monotonic's clock: range seen final / filled and counted)"""
        if not self.top_n or not len(ends):
            return
        pending = self.from_pending and self.from_list is self.to_list        # (the From column of a list against itself: filled in the same call)
        self.from_pending = self.from_pending and not pending
        _lib._pack.fill_ranges(self.to_list, idx_addr, val_addr, self.top_n, tuple(self._name_at), tuple(self._sim_at),
                               tuple(int(e) for e in ends), wait_addr, ctx_addr, first_slot, _RANGE_THREADS,
                               stamps.ctypes.data if stamps is not None else 0, self.from_col.ctypes.data if pending else 0)

    def fill(self, idx, val, row0=0):
        m = len(idx)
        if not m or not self.top_n:
            return
        idx = np.ascontiguousarray(idx, np.int32).reshape(m, self.top_n)
        val = np.ascontiguousarray(val, np.float32).reshape(m, self.top_n)
        if self.from_pending and row0 == 0 and m == self.n and self.from_list is self.to_list and _RANGE_THREADS > 1 and \
                self.top_n <= 1024 and hasattr(_lib._pack, "fill_ranges"):
            self.fill_ranges(idx.ctypes.data, val.ctypes.data, (m,), 0, 0, 0)       # (the whole frame of a list against itself: From column included)
        else:
            self.fill_raw(idx.ctypes.data, val.ctypes.data, m, row0)

    def _wrap(self):
        if self._blocks is not None:
            return _fast_frame_of_blocks(*self._blocks)
        if self.top_n and self.n < 8192 and _fast_frame_ok():
            return _fast_frame(self.from_col, self.names, self.sims)
        data = {"From": self.from_col}
        for r in range(self.top_n):
            data["To" if r == 0 else f"To_{r + 1}"] = self.names[r]
            data["Similarity" if r == 0 else f"Similarity_{r + 1}"] = self.sims[r]
        return pd.DataFrame(data, copy=False)

    def frame(self):
        if self.from_pending:           # (nobody filled the From column: a path without fill_ranges)
            self.from_col[:] = object_column(self.from_list)
            self.from_pending = False
        return self._frame if self._frame is not None else self._wrap()


def topn_to_frame(idx: np.ndarray, val: np.ndarray, from_list: List[str], to_list: List[str],
                  top_n: int, from_col: np.ndarray = None) -> pd.DataFrame:
    """(idx, score) arrays -> the reference's DataFrame (_utils.py:104-125):
    columns From, To, Similarity[, To_2, Similarity_2 ...]; scores rounded to 3
    decimals (_utils.py:70,102,143); Similarity < 0.001 -> 0.0 and To -> None.

    Every (To_r, Similarity_r) pair is filled by one pass of the CPython helper
    (_pack.fill_columns: rounding, the <0.001 rule and the prefetched gather of the
    names); without the helper the numpy twin below builds the same frame."""
    if _lib._pack is None or not isinstance(to_list, (list, tuple)):
        return _topn_to_frame_numpy(idx, val, from_list, to_list, top_n)
    # (a big list against itself -- the sharded self-match's frame: the crew that gathers the To columns fills the From column too)
    fb = FrameBuilder(from_list, to_list, top_n, from_col,
                      from_pending=from_col is None and from_list is to_list and len(from_list) * top_n >= 65536)
    fb.fill(idx, val, 0)
    return fb.frame()


def _topn_to_frame_numpy(idx, val, from_list, to_list, top_n) -> pd.DataFrame:
    """numpy twin of topn_to_frame (fallback when _pack.so is not built; the tests compare the two)."""
    n = len(from_list)
    from_arr = object_column(from_list)
    sim = np.round(np.asarray(val, np.float64).reshape(n, top_n), 3)
    j = np.asarray(idx, np.int64).reshape(n, top_n)
    none = (sim < 0.001) | (j < 0) | (j >= len(to_list))
    sim[none] = 0.0
    j = np.where(none, len(to_list), j)
    to_arr = np.empty(len(to_list) + 1, dtype=object)
    to_arr[:len(to_list)] = to_list
    to_arr[len(to_list)] = None
    data = {"From": from_arr}
    for r in range(top_n):
        data["To" if r == 0 else f"To_{r + 1}"] = to_arr[j[:, r]]
        data["Similarity" if r == 0 else f"Similarity_{r + 1}"] = sim[:, r].copy()
    return pd.DataFrame(data, copy=False)


def clip_top_n(top_n: int, to_list) -> int:
    """reference _utils.py:54-56: top_n = min(top_n, len(set(to_list))) -- without hashing the
    whole to-list on every call: stop as soon as top_n distinct strings have been seen."""
    if to_list is None or top_n <= 1 and len(to_list) >= 1:
        return top_n
    seen = set()
    for s in to_list:
        seen.add(s)
        if len(seen) >= top_n:
            return top_n
    return len(seen)


def _l2_rows(m):
    """rows / ||row||_2 (zero rows stay zero), float64 -- host preparation of caller-supplied matrices"""
    m = m.tocsr().astype(np.float64)
    norms = np.sqrt(np.asarray(m.multiply(m).sum(axis=1)).ravel())
    norms[norms == 0.0] = 1.0
    from scipy.sparse import diags
    return (diags(1.0 / norms) @ m).tocsr()


def _to_device_csr(ctx, m):
    if issparse(m):
        return _lib.DeviceCSR.from_scipy(ctx, m)
    raise TypeError(f"expected a scipy sparse matrix, got {type(m)!r}")


def cosine_similarity(from_vector,
                      to_vector,
                      from_list: List[str],
                      to_list: List[str],
                      min_similarity: float = 0.75,
                      top_n: int = 1,
                      method: str = "sparse") -> pd.DataFrame:
    """ Calculate similarity between two matrices/vectors and return best matches

    Arguments mirror reference _utils.py:15-21.  `from_vector` / `to_vector` are
    scipy sparse matrices (TF-IDF) or dense ndarrays (embeddings); `to_list=None`
    means self-match: the diagonal is excluded (_utils.py:84-87, 97-98).
    """
    if method not in _METHODS:
        raise ValueError(f"method must be one of {_METHODS}, got {method!r}")
    top_n = clip_top_n(top_n, to_list)
    self_match = to_list is None
    lower = float(min_similarity) if method in ("sparse", "hip") else 0.0
    ctx = _lib.Context.default()

    if isinstance(from_vector, np.ndarray) != isinstance(to_vector, np.ndarray):
        # one dense, one sparse operand: the reference turns the ndarray into a csr_matrix (_utils.py:74-77)
        from scipy.sparse import csr_matrix
        if isinstance(from_vector, np.ndarray):
            from_vector = csr_matrix(from_vector)
        else:
            to_vector = csr_matrix(to_vector)
    if isinstance(from_vector, np.ndarray):
        # "sparse" multiplies the arrays as they are (reference _utils.py:74-82: csr_matrix(ndarray), no
        # normalisation); "sklearn"/"knn"/"hip" are true cosines (_utils.py:59-70,94-95)
        idx, val = _lib.dense_cossim_topn_host(ctx, np.asarray(from_vector), np.asarray(to_vector), max(top_n, 1),
                                               lower, self_match, normalize=method != "sparse")
    else:
        if method in ("sklearn", "knn"):
            # true cosines whatever the caller's row norms are (sklearn pairwise.py normalises both operands;
            # a no-op for TF-IDF rows).  "sparse" multiplies the matrices as they are (_utils.py:82)
            same = to_vector is from_vector
            from_vector = _l2_rows(from_vector)
            to_vector = from_vector if same else _l2_rows(to_vector)
        a = _to_device_csr(ctx, from_vector)
        b = a if to_vector is from_vector else _to_device_csr(ctx, to_vector)
        index = _lib.DeviceIndex.build(ctx, b)
        idx, val = _lib.cossim_topn(ctx, index, a, max(top_n, 1), lower, exclude_diag=self_match).download()
    if self_match:
        to_list = from_list if isinstance(from_list, (list, tuple)) else list(from_list)
    return topn_to_frame(idx, val, from_list, to_list, top_n)
