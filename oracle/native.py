"""ctypes loader for oracle/_build/liboracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the C restatement with gcc (no GPU needed)."""
    srcs = [os.path.join(_HERE, f) for f in ("cossim_topn.c", "indel.c", "fuzz_scorers.c")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    lib = ctypes.CDLL(_SO)
    lib.oracle_cossim_topn.restype = ctypes.c_int
    lib.oracle_cossim_topn.argtypes = [
        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
        _i64p, _i32p, _f64p, _i64p, _i32p, _f64p,
        ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_double, ctypes.c_int32,
        _i32p, _f64p]
    lib.oracle_cossim_topn_rows.restype = ctypes.c_int
    lib.oracle_cossim_topn_rows.argtypes = [
        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
        _i64p, _i32p, _f64p, _i64p, _i32p, _f64p,
        _i64p, ctypes.c_int64, ctypes.c_int32, ctypes.c_double, ctypes.c_int32,
        _i32p, _f64p]
    lib.oracle_cossim_dense.restype = ctypes.c_int
    lib.oracle_cossim_dense.argtypes = [
        ctypes.c_int64, ctypes.c_int64,
        _i64p, _i32p, _f64p, _i64p, _i32p, _f64p,
        ctypes.c_int64, ctypes.c_int64, _f64p]
    lib.oracle_indel_ratio.restype = ctypes.c_double
    lib.oracle_indel_ratio.argtypes = [_u32p, ctypes.c_int64, _u32p, ctypes.c_int64]
    lib.oracle_indel_argmax.restype = ctypes.c_int
    lib.oracle_indel_argmax.argtypes = [
        _u32p, _i64p, ctypes.c_int64, _u32p, _i64p, ctypes.c_int64,
        ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
        _i32p, _f64p, ctypes.c_void_p]
    lib.oracle_fuzz_score.restype = ctypes.c_double
    lib.oracle_fuzz_score.argtypes = [_u32p, ctypes.c_int64, _u32p, ctypes.c_int64, ctypes.c_int32]
    lib.oracle_fuzz_extract_one.restype = ctypes.c_int
    lib.oracle_fuzz_extract_one.argtypes = [
        _u32p, _i64p, ctypes.c_int64, _u32p, _i64p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_int64, ctypes.c_int64, _i32p, _f64p]
    lib.oracle_fuzz_matrix.restype = ctypes.c_int
    lib.oracle_fuzz_matrix.argtypes = [_u32p, _i64p, ctypes.c_int64, _u32p, _i64p, ctypes.c_int64, ctypes.c_int32,
                                       ctypes.c_int64, ctypes.c_int64, _f64p]
    _lib = lib
    return lib


def _csr(m):
    indptr, idx, val = m
    return (np.ascontiguousarray(indptr, np.int64), np.ascontiguousarray(idx, np.int32),
            np.ascontiguousarray(val, np.float64))


def cossim_topn(a_csr, b_csr, n_col, ntop, lower_bound, exclude_diag=False, rows=None):
    """a_csr/b_csr = (indptr, indices, data).  Returns (idx int32, val float64), shape (n, ntop).
    rows: None (all), (begin, end), or an int64 ndarray of row ids (results in that order)."""
    lib = _load()
    ap, ai, av = _csr(a_csr)
    bp, bi, bv = _csr(b_csr)
    n_a, n_b = len(ap) - 1, len(bp) - 1
    if isinstance(rows, np.ndarray):                       # an arbitrary selection of rows (e.g. a seeded random sample)
        ids = np.ascontiguousarray(rows, np.int64)
        out_idx = np.empty((len(ids), ntop), np.int32)
        out_val = np.empty((len(ids), ntop), np.float64)
        rc = lib.oracle_cossim_topn_rows(n_a, n_b, n_col, ap, ai, av, bp, bi, bv, ids, len(ids), ntop, float(lower_bound),
                                         int(bool(exclude_diag)), out_idx, out_val)
        if rc != 0:
            raise ValueError(f"oracle_cossim_topn_rows failed ({rc})")
        return out_idx, out_val
    r0, r1 = (0, n_a) if rows is None else rows
    out_idx = np.empty((r1 - r0, ntop), np.int32)
    out_val = np.empty((r1 - r0, ntop), np.float64)
    rc = lib.oracle_cossim_topn(n_a, n_b, n_col, ap, ai, av, bp, bi, bv, r0, r1,
                                ntop, float(lower_bound), int(bool(exclude_diag)), out_idx, out_val)
    if rc != 0:
        raise MemoryError("oracle_cossim_topn failed")
    return out_idx, out_val


def cossim_dense(a_csr, b_csr, n_col, rows=None):
    lib = _load()
    ap, ai, av = _csr(a_csr)
    bp, bi, bv = _csr(b_csr)
    n_a, n_b = len(ap) - 1, len(bp) - 1
    r0, r1 = (0, n_a) if rows is None else rows
    out = np.empty((r1 - r0, n_b), np.float64)
    rc = lib.oracle_cossim_dense(n_b, n_col, ap, ai, av, bp, bi, bv, r0, r1, out)
    if rc != 0:
        raise MemoryError("oracle_cossim_dense failed")
    return out


def _codepoints(strings):
    lens = np.fromiter((len(s) for s in strings), np.int64, len(strings))
    off = np.zeros(len(strings) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    joined = "".join(strings)
    if joined:
        cp = np.frombuffer(joined.encode("utf-32-le", "surrogatepass"), np.uint32).copy()
    else:
        cp = np.zeros(1, np.uint32)
    return cp, off


def indel_ratio(a, b):
    lib = _load()
    ca = np.array([ord(c) for c in a] or [0], np.uint32)
    cb = np.array([ord(c) for c in b] or [0], np.uint32)
    return lib.oracle_indel_ratio(ca, len(a), cb, len(b))


def indel_argmax(from_list, to_list, self_match=False, rows=None, want_matrix=False):
    """Returns (idx int32[n], score float64[n][, matrix])."""
    lib = _load()
    acp, aoff = _codepoints(from_list)
    bcp, boff = _codepoints(to_list)
    n_a, n_b = len(from_list), len(to_list)
    r0, r1 = (0, n_a) if rows is None else rows
    out_idx = np.empty(r1 - r0, np.int32)
    out_score = np.empty(r1 - r0, np.float64)
    mat = np.empty((r1 - r0, n_b), np.float64) if want_matrix else None
    rc = lib.oracle_indel_argmax(acp, aoff, n_a, bcp, boff, n_b, r0, r1, int(bool(self_match)),
                                 out_idx, out_score,
                                 mat.ctypes.data_as(ctypes.c_void_p) if mat is not None else None)
    if rc != 0:
        raise MemoryError("oracle_indel_argmax failed")
    return (out_idx, out_score, mat) if want_matrix else (out_idx, out_score)


# the order of oracle/fuzz_scorers.c's scorer enum
FUZZ_SCORER_IDS = {name: i for i, name in enumerate(
    ("ratio", "QRatio", "partial_ratio", "token_sort_ratio", "token_set_ratio", "token_ratio", "partial_token_sort_ratio",
     "partial_token_set_ratio", "partial_token_ratio", "WRatio"))}


def fuzz_score(a, b, scorer):
    """One pair under the rapidfuzz.fuzz scorer `scorer` (oracle/fuzz_scorers.c, == oracle/fuzz_scorers.py)."""
    lib = _load()
    ca = np.array([ord(c) for c in a] or [0], np.uint32)
    cb = np.array([ord(c) for c in b] or [0], np.uint32)
    return lib.oracle_fuzz_score(ca, len(a), cb, len(b), FUZZ_SCORER_IDS[scorer])


def fuzz_extract_one(from_list, to_list, scorer, skip=None, rows=None):
    """process.extractOne(from_string, to_list, scorer=fuzz.<scorer>) for from-rows `rows` = (begin, end):
    (index of the first best choice int32 (-1: none), its score float64 on the 0..100 scale)."""
    lib = _load()
    acp, aoff = _codepoints(from_list)
    bcp, boff = _codepoints(to_list)
    r0, r1 = (0, len(from_list)) if rows is None else rows
    out_idx = np.empty(r1 - r0, np.int32)
    out_score = np.empty(r1 - r0, np.float64)
    sk = None
    if skip is not None:
        sk = np.ascontiguousarray(skip, np.int32)
        assert len(sk) == len(from_list)
    rc = lib.oracle_fuzz_extract_one(acp, aoff, len(from_list), bcp, boff, len(to_list), FUZZ_SCORER_IDS[scorer],
                                     sk.ctypes.data_as(ctypes.c_void_p) if sk is not None else None, r0, r1, out_idx, out_score)
    if rc != 0:
        raise ValueError(f"oracle_fuzz_extract_one failed ({rc})")
    return out_idx, out_score


def fuzz_matrix(from_list, to_list, scorer, rows=None):
    """float64 [rows, len(to_list)]: every pair's score under the rapidfuzz.fuzz scorer `scorer`."""
    lib = _load()
    acp, aoff = _codepoints(from_list)
    bcp, boff = _codepoints(to_list)
    r0, r1 = (0, len(from_list)) if rows is None else rows
    out = np.empty((r1 - r0, len(to_list)), np.float64)
    rc = lib.oracle_fuzz_matrix(acp, aoff, len(from_list), bcp, boff, len(to_list), FUZZ_SCORER_IDS[scorer], r0, r1, out)
    if rc != 0:
        raise ValueError(f"oracle_fuzz_matrix failed ({rc})")
    return out
