"""ctypes binding of libpolyfuzz_hip.so (the C ABI of include/polyfuzz_hip.h).

No torch, no fallback: if the library is missing or no gfx950 device is visible
the calls raise -- the host layer never computes similarities itself.
"""
import ctypes
import os
import threading

import numpy as np

from . import _build

c_i32, c_i64, c_f32, c_f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double
c_vp = ctypes.c_void_p
P = ctypes.POINTER


class PfzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libpolyfuzz_hip error {code}: {msg}")
        self.code = code


class PfzUnsupported(PfzError, NotImplementedError):
    pass


class PfzNoDevice(PfzError):
    pass


class TfidfParams(ctypes.Structure):
    _fields_ = [("ngram_lo", c_i32), ("ngram_hi", c_i32), ("clean", c_i32), ("remove_space_ngrams", c_i32)]


# every symbol include/polyfuzz_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "pfz_version": (ctypes.c_int, []),
    "pfz_last_error": (ctypes.c_char_p, []),
    "pfz_device_count": (ctypes.c_int, []),
    "pfz_ctx_create": (ctypes.c_int, [ctypes.c_int, P(c_vp)]),
    "pfz_ctx_destroy": (None, [c_vp]),
    "pfz_ctx_sync": (ctypes.c_int, [c_vp]),
    "pfz_ctx_info": (ctypes.c_int, [c_vp, ctypes.c_char_p, P(c_i32), P(c_i64)]),
    "pfz_event_record": (ctypes.c_int, [c_vp, c_i32]),
    "pfz_event_elapsed_ms": (ctypes.c_int, [c_vp, c_i32, c_i32, P(c_f32)]),
    "pfz_prof_enable": (ctypes.c_int, [c_vp, c_i32]),
    "pfz_prof_reset": (ctypes.c_int, [c_vp]),
    "pfz_prof_get": (ctypes.c_int, [c_vp, ctypes.c_char_p, P(c_f64), P(c_i64)]),
    "pfz_csr_upload": (ctypes.c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, P(c_vp)]),
    "pfz_csr_shape": (ctypes.c_int, [c_vp, P(c_i64), P(c_i64), P(c_i64)]),
    "pfz_csr_download": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pfz_csr_free": (None, [c_vp]),
    "pfz_index_build": (ctypes.c_int, [c_vp, c_vp, P(c_vp)]),
    "pfz_index_free": (None, [c_vp]),
    "pfz_index_info": (ctypes.c_int, [c_vp, P(c_i64), P(c_i64), P(c_i64), P(c_i64), P(c_i64), P(c_i64)]),
    "pfz_index_pieces": (ctypes.c_int, [c_vp, P(c_i64), P(c_i64)]),
    "pfz_index_symmetric_launches": (ctypes.c_int, [c_vp, P(c_i64), P(c_i64)]),
    "pfz_index_symmetric_census": (ctypes.c_int, [c_vp, P(c_i64), P(c_i64)]),
    "pfz_index_symmetric_ok": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, P(c_i32)]),
    "pfz_comm_cossim_topn_symmetric": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, ctypes.c_float, c_vp]),
    "pfz_comm_symmetric_ok": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, P(c_i32)]),
    "pfz_rccl_versions": (ctypes.c_int, [P(c_i32), P(c_i32)]),
    "pfz_topn_alloc": (ctypes.c_int, [c_vp, c_i64, c_i32, P(c_vp)]),
    "pfz_topn_free": (None, [c_vp]),
    "pfz_topn_download": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "pfz_topn_clear": (ctypes.c_int, [c_vp, c_vp]),
    "pfz_topn_download_rows_after": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp]),
    "pfz_topn_rows_begin": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_i32, c_i32]),
    "pfz_topn_rows_finish": (ctypes.c_int, [c_vp, c_i32, P(c_vp), P(c_vp)]),
    "pfz_cossim_topn_ranges": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, ctypes.c_float, c_i32, c_i32, P(c_i64), c_i32, c_vp, P(c_vp), P(c_vp)]),
    "pfz_event_wait": (ctypes.c_int, [c_vp, c_i32]),
    "pfz_stage_reserve": (ctypes.c_int, [c_vp, c_i64, P(c_vp)]),
    "pfz_topn_upload": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "pfz_topn_device_ptrs": (ctypes.c_int, [c_vp, P(c_vp), P(c_vp), P(c_i64), P(c_i32)]),
    "pfz_cossim_topn": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_f32, c_i32, c_i64, c_vp]),
    "pfz_cossim_topn_rows": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_f32, c_i32, c_i64, c_vp]),
    "pfz_cossim_topn_host": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                            c_i32, c_f32, c_i32, c_vp, c_vp]),
    "pfz_strings_upload": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, P(c_vp)]),
    "pfz_strings_free": (None, [c_vp]),
    "pfz_tfidf_fit": (ctypes.c_int, [c_vp, P(TfidfParams), c_vp, c_vp, P(c_vp)]),
    "pfz_tfidf_free": (None, [c_vp]),
    "pfz_tfidf_info": (ctypes.c_int, [c_vp, P(c_i64), P(c_i64), P(c_i32)]),
    "pfz_tfidf_export": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pfz_tfidf_import": (ctypes.c_int, [c_vp, P(TfidfParams), c_i64, c_i64, c_vp, c_vp, P(c_vp)]),
    "pfz_tfidf_transform": (ctypes.c_int, [c_vp, c_vp, c_vp, P(c_vp)]),
    "pfz_indel_argmax": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "pfz_indel_matrix_host": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "pfz_indel_plan_info": (ctypes.c_int, [c_vp, c_vp, P(c_i64), P(c_i64), P(c_i64)]),
    "pfz_indel_argmax_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "pfz_fuzz_extract_one": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "pfz_fuzz_extract_one_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "pfz_fuzz_plan_info": (ctypes.c_int, [c_vp, c_vp, P(c_i64), P(c_i64), P(c_i64), P(c_i64)]),
    "pfz_dense_cossim_topn_host": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_f32, c_i32,
                                                  c_vp, c_vp]),
    "pfz_dense_dot_topn_host": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_f32, c_i32,
                                               c_vp, c_vp]),
    "pfz_dense_upload": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_i32, P(c_vp)]),
    "pfz_dense_shape": (ctypes.c_int, [c_vp, P(c_i64), P(c_i64)]),
    "pfz_dense_free": (None, [c_vp]),
    "pfz_dense_topn": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_f32, c_i32, c_i64, c_vp]),
    "pfz_pr_curve_host": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp]),
    "pfz_linkage_top1": (ctypes.c_int, [c_vp, c_vp, c_f64, c_vp, c_vp, c_vp]),
    "pfz_comm_unique_id": (ctypes.c_int, [c_vp]),
    "pfz_comm_init": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, P(c_vp)]),
    "pfz_comm_destroy": (None, [c_vp]),
    "pfz_comm_group_create": (ctypes.c_int, [c_i32, P(c_vp)]),
    "pfz_comm_group_destroy": (None, [c_vp]),
    "pfz_comm_init_local": (ctypes.c_int, [c_vp, c_vp, c_i32, P(c_vp)]),
    "pfz_comm_allgather_topn": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "pfz_comm_merge_to_shards": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "pfz_comm_barrier": (ctypes.c_int, [c_vp]),
    "pfz_comm_info": (ctypes.c_int, [c_vp, P(c_i32), P(c_i32)]),
    "pfz_tfidf_fit_sharded": (ctypes.c_int, [c_vp, c_vp, P(TfidfParams), c_vp, c_vp, P(c_vp)]),
}

_lib = None
_lock = threading.Lock()


def lib_path():
    # POLYFUZZ_HIP_LIB: another build of the same sources (tuning experiments: tools/build_variant.sh)
    return os.environ.get("POLYFUZZ_HIP_LIB") or _build.LIB_PATH


def load():
    """Load libpolyfuzz_hip.so (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950).  polyfuzz_amd has no CPU fallback.")
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc):
    if rc == 0:
        return
    msg = load().pfz_last_error().decode("utf-8", "replace")
    if rc == -4:
        raise PfzUnsupported(rc, msg)
    if rc == -2:
        raise PfzNoDevice(rc, msg)
    raise PfzError(rc, msg)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_vp)


def device_count():
    return int(load().pfz_device_count())


class Context:
    """One HIP device + stream.  `Context.default()` is a per-process singleton on
    device POLYFUZZ_HIP_DEVICE (or LOCAL_RANK, or 0)."""
    _default = None

    def __init__(self, device=None):
        if device is None:
            device = int(os.environ.get("POLYFUZZ_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self.lib = load()
        h = c_vp()
        check(self.lib.pfz_ctx_create(int(device), ctypes.byref(h)))
        self.h = h
        self.device = int(device)

    @classmethod
    def default(cls):
        if cls._default is None:
            cls._default = cls()
        return cls._default

    def close(self):
        if getattr(self, "h", None):
            self.lib.pfz_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.pfz_ctx_sync(self.h))

    def info(self):
        name = ctypes.create_string_buffer(256)
        ncu, mem = c_i32(), c_i64()
        check(self.lib.pfz_ctx_info(self.h, name, ctypes.byref(ncu), ctypes.byref(mem)))
        return {"name": name.value.decode(), "n_cu": ncu.value, "hbm_bytes": mem.value}

    # timers ---------------------------------------------------------------
    def event_record(self, slot):
        check(self.lib.pfz_event_record(self.h, slot))

    def event_wait(self, slot):
        """block until event slot `slot` has fired (pfz_event_wait)"""
        check(self.lib.pfz_event_wait(self.h, slot))

    def event_wait_fn(self):
        """(address of pfz_event_wait, the context's handle as an integer): what a native consumer needs to wait for event slots
        itself, without Python in between (_pack.fill_ranges)"""
        return ctypes.cast(self.lib.pfz_event_wait, ctypes.c_void_p).value, self.h.value if hasattr(self.h, "value") else int(self.h)

    def event_elapsed_ms(self, a, b):
        ms = c_f32()
        check(self.lib.pfz_event_elapsed_ms(self.h, a, b, ctypes.byref(ms)))
        return float(ms.value)

    def prof_enable(self, on=True):
        """on: False / True (every profiled kernel) / 2 (the dominant kernels only: least overhead)"""
        check(self.lib.pfz_prof_enable(self.h, int(on)))

    def prof_reset(self):
        check(self.lib.pfz_prof_reset(self.h))

    def prof_get(self, name):
        ms, n = c_f64(), c_i64()
        check(self.lib.pfz_prof_get(self.h, name.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)


class _Handle:
    _free = None

    def __init__(self, ctx, h):
        self.ctx = ctx
        self.h = h

    def free(self):
        if getattr(self, "h", None) and self.ctx.h:
            getattr(self.ctx.lib, self._free)(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceCSR(_Handle):
    _free = "pfz_csr_free"

    @classmethod
    def upload(cls, ctx, indptr, indices, data, n_cols):
        indptr = np.ascontiguousarray(indptr, np.int64)
        indices = np.ascontiguousarray(indices, np.int32)
        data = np.ascontiguousarray(data, np.float32)
        h = c_vp()
        check(ctx.lib.pfz_csr_upload(ctx.h, len(indptr) - 1, int(n_cols), _ptr(indptr), _ptr(indices), _ptr(data),
                                     ctypes.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_scipy(cls, ctx, m):
        m = m.tocsr()
        if not m.has_sorted_indices:
            m = m.sorted_indices()
        return cls.upload(ctx, m.indptr, m.indices, m.data, m.shape[1])

    @property
    def shape(self):
        r, c, z = c_i64(), c_i64(), c_i64()
        check(self.ctx.lib.pfz_csr_shape(self.h, ctypes.byref(r), ctypes.byref(c), ctypes.byref(z)))
        return r.value, c.value, z.value

    @property
    def n_rows(self):
        """rows alone: known when the matrix is enqueued -- `shape` also asks for the count of non-zeros, which waits for the
        kernels that produce it (a transform's count travels to the host behind its last kernel)"""
        r = c_i64()
        check(self.ctx.lib.pfz_csr_shape(self.h, ctypes.byref(r), None, None))
        return r.value

    def download(self):
        n_rows, n_cols, nnz = self.shape
        indptr = np.empty(n_rows + 1, np.int64)
        indices = np.empty(nnz, np.int32)
        data = np.empty(nnz, np.float32)
        check(self.ctx.lib.pfz_csr_download(self.ctx.h, self.h, _ptr(indptr), _ptr(indices), _ptr(data)))
        return indptr, indices, data, n_cols


class DeviceIndex(_Handle):
    _free = "pfz_index_free"

    @classmethod
    def build(cls, ctx, to_csr):
        h = c_vp()
        check(ctx.lib.pfz_index_build(ctx.h, to_csr.h, ctypes.byref(h)))
        return cls(ctx, h)

    def info(self):
        v = [c_i64() for _ in range(6)]
        check(self.ctx.lib.pfz_index_info(self.h, *[ctypes.byref(x) for x in v]))
        keys = ("n_rows", "n_cols", "nnz", "block_cols", "n_blocks", "table_bytes")
        out = dict(zip(keys, (x.value for x in v)))
        npc, per = c_i64(), c_i64()
        check(self.ctx.lib.pfz_index_pieces(self.h, ctypes.byref(npc), ctypes.byref(per)))
        out.update(n_pieces=npc.value, piece_postings=per.value)
        return out

    def symmetric_launches(self):
        """(launches, from-rows) of this index that K3 served in its symmetric self-match form (k3_symmetric.hip)"""
        a, b = c_i64(), c_i64()
        check(self.ctx.lib.pfz_index_symmetric_launches(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def symmetric_ok(self, csr, ntop, n_parts=1):
        """may a self-match of `csr` (the matrix this index was built from) run in K3's symmetric form, cut over n_parts GPUs?"""
        yes = c_i32()
        check(self.ctx.lib.pfz_index_symmetric_ok(self.h, csr.h, int(ntop), int(n_parts), ctypes.byref(yes)))
        return bool(yes.value)

    def symmetric_census(self):
        """(magnet rows, rows recomputed row-major) of the last symmetric launch on this index (k3_symmetric.hip)"""
        a, b = c_i64(), c_i64()
        check(self.ctx.lib.pfz_index_symmetric_census(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value


class DeviceTopN(_Handle):
    _free = "pfz_topn_free"

    @classmethod
    def alloc(cls, ctx, n_rows, ntop):
        h = c_vp()
        check(ctx.lib.pfz_topn_alloc(ctx.h, int(n_rows), int(ntop), ctypes.byref(h)))
        t = cls(ctx, h)
        t.n_rows, t.ntop = int(n_rows), int(ntop)
        return t

    def download(self):
        idx = np.empty((self.n_rows, self.ntop), np.int32)
        val = np.empty((self.n_rows, self.ntop), np.float32)
        check(self.ctx.lib.pfz_topn_download(self.ctx.h, self.h, _ptr(idx), _ptr(val)))
        return idx, val

    def clear(self):
        check(self.ctx.lib.pfz_topn_clear(self.ctx.h, self.h))

    def download_rows_after(self, begin, end, event_slot):
        """Rows [begin, end) as soon as the context's event `event_slot` has fired (side stream; later work keeps running)."""
        idx = np.empty((end - begin, self.ntop), np.int32)
        val = np.empty((end - begin, self.ntop), np.float32)
        check(self.ctx.lib.pfz_topn_download_rows_after(self.ctx.h, self.h, int(begin), int(end), int(event_slot), _ptr(idx),
                                                        _ptr(val)))
        return idx, val

    def rows_begin(self, begin, end, event_slot, half):
        """enqueue the download of rows [begin, end) behind the context's event `event_slot` into pinned staging half 0 / 1"""
        check(self.ctx.lib.pfz_topn_rows_begin(self.ctx.h, self.h, int(begin), int(end), int(event_slot), int(half)))

    def rows_finish(self, half):
        """wait for the download begun on `half`: (address of int32 idx[rows][ntop], address of fp32 val[rows][ntop]) in PINNED
        memory of the context, valid until the next rows_begin on that half"""
        a, b = c_vp(), c_vp()
        check(self.ctx.lib.pfz_topn_rows_finish(self.ctx.h, int(half), ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    @classmethod
    def from_host(cls, ctx, idx, val):
        idx = np.ascontiguousarray(idx, np.int32)
        val = np.ascontiguousarray(val, np.float32)
        if idx.ndim == 1:
            idx, val = idx[:, None], val[:, None]
        t = cls.alloc(ctx, idx.shape[0], idx.shape[1])
        check(ctx.lib.pfz_topn_upload(ctx.h, t.h, _ptr(idx), _ptr(val)))
        return t

    def upload(self, idx, val):
        """overwrite the whole buffer with host arrays of its shape"""
        idx = np.ascontiguousarray(idx, np.int32).reshape(self.n_rows, self.ntop)
        val = np.ascontiguousarray(val, np.float32).reshape(self.n_rows, self.ntop)
        check(self.ctx.lib.pfz_topn_upload(self.ctx.h, self.h, _ptr(idx), _ptr(val)))

    def device_ptrs(self):
        pi, pv = c_vp(), c_vp()
        check(self.ctx.lib.pfz_topn_device_ptrs(self.h, ctypes.byref(pi), ctypes.byref(pv), None, None))
        return pi.value, pv.value


def cossim_topn(ctx, index, from_csr, ntop, lower_bound, exclude_diag=False, diag_offset=0, out=None, rows=None):
    """Enqueue K3 (for from-rows `rows` = (begin, end) only, if given); returns the (device-resident) DeviceTopN."""
    if out is None:
        out = DeviceTopN.alloc(ctx, from_csr.n_rows, ntop)      # (not .shape: that would wait for the transform)
    if rows is None:
        check(ctx.lib.pfz_cossim_topn(ctx.h, index.h, from_csr.h, int(ntop), float(lower_bound),
                                      int(bool(exclude_diag)), int(diag_offset), out.h))
    else:
        check(ctx.lib.pfz_cossim_topn_rows(ctx.h, index.h, from_csr.h, int(rows[0]), int(rows[1]), int(ntop),
                                           float(lower_bound), int(bool(exclude_diag)), int(diag_offset), out.h))
    return out


def cossim_topn_ranges(ctx, index, from_csr, ntop, lower_bound, exclude_diag, range_ends, first_event, out=None, mirror=False):
    """Enqueue the whole match with its results handed on in ascending row ranges (pfz_cossim_topn_ranges): range i ends at
    range_ends[i], the context's event first_event + i fires when its rows are final.  Returns the DeviceTopN -- with mirror=True:
    (DeviceTopN, idx address, val address), the addresses of int32 idx[n][ntop] / fp32 val[n][ntop] in pinned HOST memory that
    the device fills beside the result (rows of range i valid after ctx.event_wait(first_event + i)), or (DeviceTopN, None,
    None) where the job does not run in the form that writes one."""
    if out is None:
        out = DeviceTopN.alloc(ctx, from_csr.n_rows, ntop)
    ends = (c_i64 * len(range_ends))(*[int(e) for e in range_ends])
    hi, hv = c_vp(), c_vp()
    check(ctx.lib.pfz_cossim_topn_ranges(ctx.h, index.h, from_csr.h, int(ntop), float(lower_bound), int(bool(exclude_diag)),
                                         len(range_ends), ends, int(first_event), out.h,
                                         ctypes.byref(hi) if mirror else None, ctypes.byref(hv) if mirror else None))
    return (out, hi.value, hv.value) if mirror else out


def cossim_topn_host(ctx, from_csr3, to_csr3, n_cols, ntop, lower_bound, exclude_diag=False):
    """One-shot: (indptr, indices, data) host triples in, (idx, val) host arrays out."""
    fp, fi, fv = from_csr3
    tp, ti, tv = to_csr3
    fp = np.ascontiguousarray(fp, np.int64)
    fi = np.ascontiguousarray(fi, np.int32)
    fv = np.ascontiguousarray(fv, np.float32)
    tp = np.ascontiguousarray(tp, np.int64)
    ti = np.ascontiguousarray(ti, np.int32)
    tv = np.ascontiguousarray(tv, np.float32)
    n_from, n_to = len(fp) - 1, len(tp) - 1
    idx = np.empty((n_from, ntop), np.int32)
    val = np.empty((n_from, ntop), np.float32)
    check(ctx.lib.pfz_cossim_topn_host(ctx.h, n_from, n_to, int(n_cols), _ptr(fp), _ptr(fi), _ptr(fv),
                                       _ptr(tp), _ptr(ti), _ptr(tv), int(ntop), float(lower_bound),
                                       int(bool(exclude_diag)), _ptr(idx), _ptr(val)))
    return idx, val


# ---- strings / vectoriser -------------------------------------------------------

try:  # CPython helper built by _build.build_host_helpers(); same result as the Python code below
    from . import _pack
except ImportError:  # pragma: no cover - not built
    _pack = None


def usable_cpus():
    """CPUs this process may really use: its affinity mask, capped by the cgroup's CPU quota where there is one (the MI355X boxes
    show 256 CPUs and grant 16 CPUs' worth of time)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, int(quota) // int(period))
    except (OSError, ValueError):
        pass
    return max(1, n)


def host_threads():
    """Host threads of the two big host-side walks of a match -- the string packer (pack_into / pack) and the frame's gathers
    (_pack.fill_ranges) -- the calling thread included.  PFZ_HOST_THREADS overrides (1 = everything on the calling thread); by
    default eight -- a CCD of the MI355X host -- where the process may use twice that many CPUs, half its CPUs below that.  The helpers run on the cores that share the calling
    thread's L3 and nowhere else (_pack.c: the crew; PFZ_HOST_PIN=0 lets the scheduler place them -- measured 4 x slower on the
    two-socket MI355X host); a host that does not tell its cache topology gets none.
    Measured there (tools/r6_match_ab.py, one box): TFIDF(top_n=5).match(100 000 names) 3.31 ms on the calling thread alone,
    3.20 / 3.02 / 2.82 / 2.79 with crews of 2 / 4 / 6 / 8."""
    env = os.environ.get("PFZ_HOST_THREADS")
    return max(1, min(16, int(env))) if env else max(1, min(8, usable_cpus() // 2))


_PACK_THREADS = _PACK_INTO_THREADS = host_threads()       # (the general packer's walks / pack_into's: separate names for the A/B tools)


def pack_strings(strings, objects=None):
    """list[str] -> (code units ndarray, offsets int64[n+1], char_width).

    1-byte code units (Latin-1) when every code point is <= 0xFF, UTF-32 otherwise.
    objects: a fresh np.empty(len(strings), object) array that is to become the From column of the result frame -- filled
    in the same walk over the strings (slot i = strings[i])."""
    if _pack is not None and isinstance(strings, (list, tuple)):
        if objects is not None:
            raw, off, width = _pack.pack(strings, _PACK_THREADS, objects.ctypes.data)
        else:
            raw, off, width = _pack.pack(strings, _PACK_THREADS)
        return np.frombuffer(raw, np.uint8 if width == 1 else np.uint32), np.frombuffer(off, np.int64), width
    if objects is not None:
        objects[:] = strings
    return _pack_strings_py(strings)


def _pack_strings_py(strings):
    n = len(strings)
    lens = np.fromiter(map(len, strings), np.int64, n)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    joined = "".join(strings)
    try:
        raw = joined.encode("latin-1")
        chars = np.frombuffer(raw, np.uint8)
        width = 1
    except UnicodeEncodeError:
        raw = joined.encode("utf-32-le", "surrogatepass")
        chars = np.frombuffer(raw, np.uint32)
        width = 4
    if len(chars) != off[-1]:
        raise ValueError("string packing length mismatch")
    return chars, off, width


class DeviceStrings(_Handle):
    _free = "pfz_strings_free"

    @classmethod
    def upload(cls, ctx, strings):
        if len(strings) >= 1024:          # (1-byte strings: packed straight into the pinned staging buffer)
            direct = cls.upload_direct(ctx, strings)
            if direct is not None:
                return direct
        return cls.upload_packed(ctx, *pack_strings(strings))

    @classmethod
    def upload_direct(cls, ctx, strings, objects=None):
        """A list of 1-byte (Latin-1) str packed STRAIGHT into the context's pinned staging buffer and uploaded from there (no
        bytes object, no copy into the staging buffer: 3 MB of host copies less in front of the device's first kernel for 100 000
        names).  objects: see pack_strings.  Returns None -- nothing changed -- when the list is not of that kind (wide or
        non-str items, too big for the staging buffer, no C helper): the caller takes pack_strings + upload_packed."""
        if _pack is None or not hasattr(_pack, "pack_into") or not isinstance(strings, (list, tuple)):
            return None
        n = len(strings)
        off_bytes = (8 * (n + 1) + 255) & ~255
        cap = off_bytes + 48 * n + 65536
        host = c_vp()
        check(ctx.lib.pfz_stage_reserve(ctx.h, cap, ctypes.byref(host)))
        if not host.value:
            return None
        n_chars = _pack.pack_into(strings, objects.ctypes.data if objects is not None else 0, host.value, off_bytes, cap, _PACK_INTO_THREADS)
        if n_chars is None:
            return None
        h = c_vp()
        check(ctx.lib.pfz_strings_upload(ctx.h, c_vp(host.value + off_bytes), c_vp(host.value), n, 1, ctypes.byref(h)))
        s = cls(ctx, h)
        s.n = n
        s.char_width = 1
        return s

    @classmethod
    def upload_packed(cls, ctx, chars, off, width):
        """Upload the result of pack_strings()."""
        n = len(off) - 1
        h = c_vp()
        check(ctx.lib.pfz_strings_upload(ctx.h, _ptr(chars) if len(chars) else None, _ptr(off), n, width,
                                         ctypes.byref(h)))
        s = cls(ctx, h)
        s.n = n
        s.char_width = width
        return s


class DeviceTfidf(_Handle):
    _free = "pfz_tfidf_free"

    @classmethod
    def fit(cls, ctx, params, docs_a, docs_b=None):
        h = c_vp()
        rc = ctx.lib.pfz_tfidf_fit(ctx.h, ctypes.byref(params), docs_a.h if docs_a is not None else None,
                                   docs_b.h if docs_b is not None else None, ctypes.byref(h))
        if rc == -1 and b"empty vocabulary" in ctx.lib.pfz_last_error():
            raise ValueError("empty vocabulary; perhaps the documents only contain stop words")
        check(rc)
        v = cls(ctx, h)
        v.params = params
        return v

    @classmethod
    def from_state(cls, ctx, params, ngrams, idf, n_docs):
        ngrams = np.ascontiguousarray(ngrams, np.uint32)
        idf = np.ascontiguousarray(idf, np.float64)
        h = c_vp()
        check(ctx.lib.pfz_tfidf_import(ctx.h, ctypes.byref(params), len(idf), int(n_docs), _ptr(ngrams), _ptr(idf),
                                       ctypes.byref(h)))
        v = cls(ctx, h)
        v.params = params
        return v

    def info(self):
        vs, nd, cb = c_i64(), c_i64(), c_i32()
        check(self.ctx.lib.pfz_tfidf_info(self.h, ctypes.byref(vs), ctypes.byref(nd), ctypes.byref(cb)))
        return {"vocab": vs.value, "n_docs": nd.value, "code_bits": cb.value}

    def export(self):
        """-> (ngrams uint32[vocab, ngram_hi], idf float64[vocab], df int64[vocab])"""
        i = self.info()
        hi = self.params.ngram_hi
        ngrams = np.empty((i["vocab"], hi), np.uint32)
        idf = np.empty(i["vocab"], np.float64)
        df = np.empty(i["vocab"], np.int64)
        check(self.ctx.lib.pfz_tfidf_export(self.ctx.h, self.h, _ptr(ngrams), _ptr(idf), _ptr(df)))
        return ngrams, idf, df

    def transform(self, docs):
        h = c_vp()
        check(self.ctx.lib.pfz_tfidf_transform(self.ctx.h, self.h, docs.h, ctypes.byref(h)))
        return DeviceCSR(self.ctx, h)


# ---- edit distance / dense ---------------------------------------------------------

def indel_argmax(ctx, from_dev, to_dev, skip_idx=None, begin=0, end=None):
    """K4: (first arg-max index int32[n], ratio float64[n]) of from-rows [begin, end)."""
    end = from_dev.n if end is None else end
    n = end - begin
    idx = np.empty(n, np.int32)
    score = np.empty(n, np.float64)
    if skip_idx is not None:
        skip_idx = np.ascontiguousarray(skip_idx, np.int32)
        if len(skip_idx) != from_dev.n:
            raise ValueError("skip_idx must have one entry per from-string")
    check(ctx.lib.pfz_indel_argmax(ctx.h, from_dev.h, to_dev.h, _ptr(skip_idx), int(begin), int(end),
                                   _ptr(idx), _ptr(score)))
    return idx, score


def indel_plan_info(ctx, to_dev):
    """Build (once) and describe K4's cached to-side plan of a DeviceStrings handle."""
    v = [c_i64() for _ in range(3)]
    check(ctx.lib.pfz_indel_plan_info(ctx.h, to_dev.h, *[ctypes.byref(x) for x in v]))
    return {"n_symbols": v[0].value, "n_groups": v[1].value, "char_steps": v[2].value}


FUZZ_SCORERS = {"WRatio": 0, "partial_ratio": 1, "token_set_ratio": 2, "token_ratio": 3, "partial_token_sort_ratio": 4,
                "partial_token_set_ratio": 5, "partial_token_ratio": 6}


def _skip_array(skip_idx, n):
    if skip_idx is None:
        return None
    skip_idx = np.ascontiguousarray(skip_idx, np.int32)
    if len(skip_idx) != n:
        raise ValueError("skip_idx must have one entry per from-string")
    return skip_idx


def _as_device_strings(ctx, strings):
    return strings if isinstance(strings, DeviceStrings) else DeviceStrings.upload(ctx, strings)


def fuzz_extract_one(ctx, from_list, to_list, scorer, skip_idx=None, begin=0, end=None):
    """K7: process.extractOne(from_string, to_list, scorer=fuzz.<scorer>) for every from-string of rows [begin, end) --
    (index of the first best choice int32[n] (-1: none), its score float64[n] on rapidfuzz's 0..100 scale).
    from_list / to_list: lists of str (uploaded here) or resident DeviceStrings -- the token forms of a list and the
    plan of a to-list are built on the device on first use and stay cached on the handle; every pair is bounded, the
    promising ones scored, on the device."""
    same = to_list is from_list
    f_dev = _as_device_strings(ctx, from_list)
    t_dev = f_dev if same else _as_device_strings(ctx, to_list)
    end = f_dev.n if end is None else end
    n = end - begin
    idx = np.empty(n, np.int32)
    score = np.empty(n, np.float64)
    skip_idx = _skip_array(skip_idx, f_dev.n)
    check(ctx.lib.pfz_fuzz_extract_one(ctx.h, f_dev.h, t_dev.h, FUZZ_SCORERS[scorer], _ptr(skip_idx), int(begin), int(end),
                                       _ptr(idx), _ptr(score)))
    return idx, score


def fuzz_extract_one_dev(ctx, from_dev, to_dev, scorer, out, skip_idx=None, begin=0, end=None, counters=False):
    """K7 with the (index, float64 score) rows left in the 2-column DeviceTopN `out` (see best_from_topn); returns the work
    counters {bounded, scored, word_steps} when asked (that waits for the kernel)."""
    end = from_dev.n if end is None else end
    skip_idx = _skip_array(skip_idx, from_dev.n)
    work = np.zeros(4, np.uint64) if counters else None
    check(ctx.lib.pfz_fuzz_extract_one_dev(ctx.h, from_dev.h, to_dev.h, FUZZ_SCORERS[scorer], _ptr(skip_idx), int(begin), int(end),
                                           out.h, _ptr(work)))
    if counters:
        return {"pairs_bounded": int(work[0]), "pairs_scored": int(work[1]), "word_steps_scored": int(work[2])}
    return None


def indel_argmax_dev(ctx, from_dev, to_dev, out, skip_idx=None, begin=0, end=None):
    """K4 with the (first arg-max, float64 ratio) rows left in the 2-column DeviceTopN `out` (see best_from_topn)."""
    end = from_dev.n if end is None else end
    skip_idx = _skip_array(skip_idx, from_dev.n)
    check(ctx.lib.pfz_indel_argmax_dev(ctx.h, from_dev.h, to_dev.h, _ptr(skip_idx), int(begin), int(end), out.h))


def best_from_topn(idx2, val2):
    """(index int32[n], score float64[n]) out of the downloaded arrays of a 2-column result buffer: column 0 of idx, the
    two fp32 value lanes of a row read as one float64"""
    return idx2[:, 0].copy(), np.ascontiguousarray(val2).view(np.float64).reshape(-1)


def fuzz_plan_info(ctx, to_dev):
    """Build (once) and describe K7's cached plan of a DeviceStrings handle used as a to-list."""
    v = [c_i64() for _ in range(4)]
    check(ctx.lib.pfz_fuzz_plan_info(ctx.h, to_dev.h, *[ctypes.byref(x) for x in v]))
    return {"n_symbols": v[0].value, "n_groups": v[1].value, "n_tokens": v[2].value, "n_general_strings": v[3].value}


def indel_matrix(ctx, from_dev, to_dev, begin=0, end=None):
    end = from_dev.n if end is None else end
    out = np.empty((end - begin, to_dev.n), np.float64)
    check(ctx.lib.pfz_indel_matrix_host(ctx.h, from_dev.h, to_dev.h, int(begin), int(end), _ptr(out)))
    return out


def dense_cossim_topn_host(ctx, from_vec, to_vec, ntop, lower_bound, exclude_diag=False, normalize=True):
    """normalize=False: raw dot products (the reference's "sparse" back-end on dense input)."""
    a = np.ascontiguousarray(from_vec, np.float32)
    b = np.ascontiguousarray(to_vec, np.float32)
    if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1]:
        raise ValueError(f"dense cosine needs two 2-D arrays with equal width, got {a.shape} and {b.shape}")
    idx = np.empty((a.shape[0], ntop), np.int32)
    val = np.empty((a.shape[0], ntop), np.float32)
    fn = ctx.lib.pfz_dense_cossim_topn_host if normalize else ctx.lib.pfz_dense_dot_topn_host
    check(fn(ctx.h, _ptr(a), a.shape[0], _ptr(b), b.shape[0], a.shape[1], int(ntop), float(lower_bound),
             int(bool(exclude_diag)), _ptr(idx), _ptr(val)))
    return idx, val


class DeviceDense(_Handle):
    """Device-resident row-major fp32 matrix + inverse row norms (K5 operand)."""
    _free = "pfz_dense_free"

    @classmethod
    def upload(cls, ctx, vec, normalize=True):
        a = np.ascontiguousarray(vec, np.float32)
        if a.ndim != 2:
            raise ValueError(f"dense vectors must be a 2-D array, got shape {a.shape}")
        h = c_vp()
        check(ctx.lib.pfz_dense_upload(ctx.h, _ptr(a) if a.size else None, a.shape[0], max(a.shape[1], 1), int(bool(normalize)),
                                       ctypes.byref(h)))
        m = cls(ctx, h)
        m.n, m.dim, m.normalize = a.shape[0], a.shape[1], bool(normalize)
        return m


def dense_topn(ctx, from_dev, to_dev, ntop, lower_bound, exclude_diag=False, diag_offset=0, out=None):
    """Enqueue K5 on resident operands; returns the (device-resident) DeviceTopN."""
    if out is None:
        out = DeviceTopN.alloc(ctx, from_dev.n, ntop)
    check(ctx.lib.pfz_dense_topn(ctx.h, from_dev.h, to_dev.h, int(ntop), float(lower_bound), int(bool(exclude_diag)),
                                 int(diag_offset), out.h))
    return out


def pr_curve(ctx, sims, thresholds):
    """K6: (count of sims >= p_k int64[k], sum of those sims float64[k]) for ascending thresholds p_k."""
    sims = np.ascontiguousarray(sims, np.float64)
    thresholds = np.ascontiguousarray(thresholds, np.float64)
    count = np.empty(len(thresholds), np.int64)
    ssum = np.empty(len(thresholds), np.float64)
    check(ctx.lib.pfz_pr_curve_host(ctx.h, _ptr(sims) if len(sims) else None, len(sims), _ptr(thresholds),
                                    len(thresholds), _ptr(count), _ptr(ssum)))
    return count, ssum


def linkage_top1(ctx, result, min_similarity):
    """K6: (cluster int32[n], key int32[n], info) of the reference's single_linkage on a self-match top-1 result."""
    cluster = np.empty(result.n_rows, np.int32)
    key = np.empty(result.n_rows, np.int32)
    info = np.zeros(3, np.int32)
    check(ctx.lib.pfz_linkage_top1(ctx.h, result.h, float(min_similarity), _ptr(cluster), _ptr(key), _ptr(info)))
    return cluster, key, {"first_row": int(info[0]), "rounds": int(info[1]), "clusters": int(info[2])}


# ---- multi-GPU (one process per GPU, RCCL) -------------------------------------------

class _LocalGroup:
    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if self.h:
                load().pfz_comm_group_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Comm(_Handle):
    """RCCL communicator bound to a Context.  Bootstrap: rank 0 creates the 128-byte
    unique id, the launcher broadcasts it (torch.distributed / MPI / file)."""
    _free = "pfz_comm_destroy"

    @staticmethod
    def unique_id():
        buf = (ctypes.c_uint8 * 128)()
        check(load().pfz_comm_unique_id(ctypes.cast(buf, c_vp)))
        return bytes(buf)

    @classmethod
    def init(cls, ctx, uid, rank, world):
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(uid)
        h = c_vp()
        check(ctx.lib.pfz_comm_init(ctx.h, ctypes.cast(buf, c_vp), int(rank), int(world), ctypes.byref(h)))
        c = cls(ctx, h)
        c.rank, c.world = int(rank), int(world)
        return c

    @classmethod
    def local_group(cls, contexts):
        """One communicator per context, all in THIS process (contexts on different GPUs or on the same one):
        host rendezvous + device-to-device copies instead of RCCL.  Each rank must be driven by its own host
        thread (the collectives block until every rank has arrived)."""
        world = len(contexts)
        g = c_vp()
        check(load().pfz_comm_group_create(world, ctypes.byref(g)))
        group = _LocalGroup(g)
        comms = []
        for rank, ctx in enumerate(contexts):
            h = c_vp()
            check(ctx.lib.pfz_comm_init_local(ctx.h, g, rank, ctypes.byref(h)))
            c = cls(ctx, h)
            c.rank, c.world, c._group = rank, world, group     # keeps the group alive
            comms.append(c)
        return comms

    @classmethod
    def from_torch_distributed(cls, ctx, dist):
        """Bootstrap through an initialised torch.distributed process group (rendezvous only)."""
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls.init(ctx, box[0], rank, world)

    def barrier(self):
        check(self.ctx.lib.pfz_comm_barrier(self.h))

    def merge_to_shards(self, local, to_offset, out=None):
        """All-gather the ranks' per-to-shard candidates and keep, on every rank, the global top-n."""
        if out is None:
            out = DeviceTopN.alloc(self.ctx, local.n_rows, local.ntop)
        check(self.ctx.lib.pfz_comm_merge_to_shards(self.h, local.h, int(to_offset), out.h))
        return out

    def symmetric_ok(self, index, csr, ntop):
        """COLLECTIVE (every rank calls it): do all ranks agree that this self-match takes K3's symmetric form, and has every rank
        the session buffers for it?  (pfz_comm_symmetric_ok)"""
        yes = c_i32(0)
        check(self.ctx.lib.pfz_comm_symmetric_ok(self.h, index.h, csr.h, int(ntop), ctypes.byref(yes)))
        return bool(yes.value)

    def cossim_topn_symmetric(self, index, csr, ntop, lower_bound, out=None):
        """the self-match of the whole (replicated) list, cut over the ranks in K3's symmetric form: the FULL result on every rank"""
        if out is None:
            out = DeviceTopN.alloc(self.ctx, csr.n_rows, ntop)
        check(self.ctx.lib.pfz_comm_cossim_topn_symmetric(self.h, index.h, csr.h, int(ntop), float(lower_bound), out.h))
        return out

    def allgather_topn(self, local, out=None):
        if out is None:
            out = DeviceTopN.alloc(self.ctx, local.n_rows * self.world, local.ntop)
        check(self.ctx.lib.pfz_comm_allgather_topn(self.h, local.h, out.h))
        return out


def rccl_versions():
    """{'header': NCCL_VERSION_CODE the library was compiled against, 'runtime': ncclGetVersion() of the librccl this process
    resolved, 'path': that library's file (from /proc/self/maps)}"""
    lib = load()
    h, r = c_i32(0), c_i32(0)
    check(lib.pfz_rccl_versions(ctypes.byref(h), ctypes.byref(r)))
    path = None
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    path = line.split()[-1]
                    break
    except OSError:
        pass
    return {"header": h.value, "runtime": r.value, "path": path}


def tfidf_fit_sharded(ctx, comm, params, replicated, local_shard):
    h = c_vp()
    rc = ctx.lib.pfz_tfidf_fit_sharded(ctx.h, comm.h, ctypes.byref(params),
                                       replicated.h if replicated is not None else None,
                                       local_shard.h if local_shard is not None else None, ctypes.byref(h))
    if rc == -1 and b"empty vocabulary" in ctx.lib.pfz_last_error():
        raise ValueError("empty vocabulary; perhaps the documents only contain stop words")
    check(rc)
    v = DeviceTfidf(ctx, h)
    v.params = params
    return v
