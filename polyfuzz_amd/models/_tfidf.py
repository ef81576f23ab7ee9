"""TFIDF -- the reference's character-n-gram TF-IDF matcher
(polyfuzz/models/_tfidf.py:15-146) on the MI355X engine.

Same constructor, same `.match(from_list, to_list=None, re_train=True)`, same
DataFrame contract.  Everything between receiving the lists and assembling the
DataFrame -- cleaning, n-gram extraction, vocabulary, tf/idf, L2 normalisation
(K1/K2), the to-side inverted index and the fused cosine top-n (K3) -- runs on
the device through the C ABI; the fitted vocabulary, idf and to-side index stay
resident in HBM between `match(..., re_train=False)` calls (reference
polyfuzz.py:234-240, "production" path).
"""
import os
import re
import time
from typing import List, Tuple

import numpy as np
import pandas as pd
from scipy.sparse import csr_matrix

from .. import _lib
from ._base import BaseMatcher
from ._utils import topn_to_frame, object_column, clip_top_n, FrameBuilder, _METHODS

_SPLIT_MIN_ROWS = 20000      # from-rows from which match() has its result handed on in row ranges, the frame built under the device's work
_SPLIT_EVENT = 48            # context event slots 48 .. 63: range i final
_SPLIT_BLOCK = 2048          # range ends are multiples of K3's to-block (what lets a self-match run ONE pass-1 launch: pfz_cossim_topn_ranges)
_SYMMETRIC_ROWS = (20480, 250000)       # the list sizes K3 takes in its symmetric form (pfz_cossim_topn, include/polyfuzz_hip.h)


def _split_ends(n, self_match=False):
    """The ends of the row ranges of a big match (PFZ_MATCH_SHARES=0.4,0.3,0.2,0.1 overrides the shares: tuning).
    Two lists: the device's time per row is flat and every range is a launch of its own -- few ranges, the last one (whose columns
    are built after the device has finished) the smallest (profiles/experiments/r04_match_split_probe.txt).
    A list against itself (K3's symmetric form, csrc/k3_symmetric.hip): a row only walks the to-blocks from its own upwards, so the
    device's time per row FALLS along the list -- 1 - (1 - x)^2 of pass 1 is spent when a share x of the rows is done -- while the
    host's time per row (the frame's gathers) is flat: the host cannot start before the first range is final and falls behind in the
    cheap last third.  Ranges cost little there (one pass-1 launch; a wait, a merge and an overflow pass per range on a side
    stream), so: twelve ranges of equal ROWS (measured on one box, tools/r6_match_ab.py: the round-5 form in four launches 3.75 ms,
    five ranges 3.6, ten 3.475, twelve 3.46, sixteen 3.48)."""
    env = os.environ.get("PFZ_MATCH_SHARES")
    shares = None
    if env:
        sh = tuple(float(x) for x in env.split(","))
        if 1 <= len(sh) <= 16 and all(x > 0 for x in sh):
            shares = tuple(x / sum(sh) for x in sh)
    if shares is None:
        if self_match and _SYMMETRIC_ROWS[0] <= n <= _SYMMETRIC_ROWS[1] and n >= 2 * _SPLIT_MIN_ROWS and os.environ.get("PFZ_K3_SYM") != "0":
            shares = (1.0 / 12,) * 12
        else:
            shares = (0.4, 0.3, 0.2, 0.1) if n >= 2 * _SPLIT_MIN_ROWS else (0.6, 0.4)
    ends, acc = [], 0.0
    for f in shares[:-1]:
        acc += f
        e = int(n * acc) // _SPLIT_BLOCK * _SPLIT_BLOCK
        if e > (ends[-1] if ends else 0) and e < n:
            ends.append(e)
    ends.append(n)
    return ends


def _clean_string(string: str) -> str:
    """ Only keep alphanumerical characters (reference _tfidf.py:142-146); host
    path for strings with code points > 0xFF, where str.lower() can map into ASCII """
    string = re.sub(r'[^A-Za-z0-9 ]+', '', string.lower())
    string = re.sub(r'\s+', ' ', string).strip()
    return string


_TRACE = bool(os.environ.get("PFZ_MATCH_TRACE"))      # host-side stamps of a big match's ranges in .last_trace (tools/r6_match_ab.py)
_DIRECT_PACK = os.environ.get("PFZ_DIRECT_PACK", "1") != "0"    # (A/B knob: 0 = pack into a bytes object, copy that into the staging buffer)
_RANGE_FILL = os.environ.get("PFZ_RANGE_FILL", "1") != "0"      # (A/B knob: 0 = the ranges waited for and filled one by one from Python, round 6's first form)
_FROM_IN_PACK = os.environ.get("PFZ_FROM_IN_PACK", "1") != "0"      # (A/B knob of tools/match_wall_probe.py; the frames are the same)


class HipTfidfVectorizer:
    """What `TFIDF.vectorizer` holds after a fit: the device-resident vocabulary +
    idf, with the read-only parts of sklearn's TfidfVectorizer surface
    (vocabulary_, idf_, get_feature_names_out, transform)."""

    def __init__(self, owner, dev):
        self._owner = owner
        self._dev = dev
        self._state = None

    def _export(self):
        if self._state is None:
            ngrams, idf, df = self._dev.export()
            names = ["".join(chr(c) for c in row if c) for row in ngrams.tolist()]
            self._state = (ngrams, idf, df, names)
        return self._state

    @property
    def idf_(self):
        return self._export()[1]

    @property
    def vocabulary_(self):
        return {g: i for i, g in enumerate(self._export()[3])}

    def get_feature_names_out(self):
        return np.array(self._export()[3], dtype=object)

    def transform(self, raw_documents) -> csr_matrix:
        docs = self._owner._upload(list(raw_documents))
        return _download_csr(self._dev.transform(docs))


def _download_csr(dev_csr) -> csr_matrix:
    indptr, indices, data, n_cols = dev_csr.download()
    return csr_matrix((data.astype(np.float64), indices, indptr), shape=(len(indptr) - 1, n_cols))


class TFIDF(BaseMatcher):
    """
    A character based n-gram TF-IDF to approximate edit distance

    Arguments (reference _tfidf.py:17-39, unchanged):
        n_gram_range: The n_gram_range on a character-level
        clean_string: Whether to clean the string such that only alphanumerical characters are kept
        min_similarity: The minimum similarity between strings, otherwise return 0 similarity
        top_n: The number of matches you want returned
        cosine_method: "sparse" (default; honours min_similarity), "sklearn", "knn" (ignore it, as the
                       reference does) -- all run the HIP kernel
        model_id: The name of the particular instance, used when comparing models
        remove_space_ngrams: Remove n-grams that contain a space

    Limits of the device path (loud `PfzUnsupported`, there is no CPU fallback): (top_n: any; beyond 128 with a larger, slower candidate buffer, beyond 1024 in passes of 1024); n-grams whose
    code (n x bits per alphabet symbol) exceeds 64 bits, e.g. 11-grams of cleaned text; fewer than 2^28
    n-gram occurrences in the to-list.
    """
    def __init__(self,
                 n_gram_range: Tuple[int, int] = (3, 3),
                 clean_string: bool = True,
                 min_similarity: float = 0.75,
                 top_n: int = 1,
                 cosine_method: str = "sparse",
                 model_id: str = None,
                 remove_space_ngrams=True):
        super().__init__(model_id)
        self.type = "TF-IDF"
        self.n_gram_range = n_gram_range
        self.clean_string = clean_string
        self.min_similarity = min_similarity
        self.cosine_method = cosine_method
        self.top_n = top_n
        self.remove_space_ngrams = remove_space_ngrams
        self._dev_vec = None        # _lib.DeviceTfidf
        self._dev_to = None         # _lib.DeviceCSR of the to-side
        self._dev_index = None      # _lib.DeviceIndex of the to-side
        self._vectorizer = None
        self._host_state = None     # picklable copy (see __getstate__)
        self.last_timings = None

    # ---- reference attributes ------------------------------------------------
    @property
    def vectorizer(self):
        self._restore()
        return self._vectorizer

    @property
    def tf_idf_to(self):
        self._restore()
        return None if self._dev_to is None else _download_csr(self._dev_to)

    # ---- API -------------------------------------------------------------------
    def match(self,
              from_list: List[str],
              to_list: List[str] = None,
              re_train: bool = True) -> pd.DataFrame:
        """ Match two lists of strings to each other and return the most similar strings
        (reference _tfidf.py:68-100) """
        if self.cosine_method not in _METHODS:
            raise ValueError(f"cosine_method must be one of {_METHODS}")
        ctx = _lib.Context.default()
        t0 = time.perf_counter()
        # the From column of the frame is filled by the string packer's own walk over from_list (a list): no second pass
        # (a big list against itself whose packer runs on threads, which touch no reference count: the From column is filled by the
        # frame's range fill, _pack.fill_ranges, while its helpers wait for the first range)
        ranged = _RANGE_FILL and _lib._PACK_INTO_THREADS > 1 and to_list is None and isinstance(from_list, list) and len(from_list) >= _SPLIT_MIN_ROWS and \
            _lib._pack is not None and hasattr(_lib._pack, "fill_ranges")
        col = [np.empty(len(from_list), dtype=object)] if isinstance(from_list, list) and len(from_list) >= 1024 and _FROM_IN_PACK and not ranged else None
        from_dev, to_dev = self._extract_tf_idf(from_list, to_list, re_train, col)
        top_n = clip_top_n(self.top_n, to_list)                   # _utils.py:54-56
        self_match = to_list is None
        lower = float(self.min_similarity) if self.cosine_method in ("sparse", "hip") else 0.0
        n = len(from_list)
        names = from_list if self_match else to_list
        # A big match has its result handed on in ascending row ranges (pfz_cossim_topn_ranges): each range is downloaded on a side
        # stream as soon as it is final and turned into frame columns while the device works on the later ones
        split = n >= _SPLIT_MIN_ROWS and top_n >= 1 and _lib._pack is not None and isinstance(names, (list, tuple))
        if split:
            ends = _split_ends(n, self_match)
            # (mirror: where the device can fill a copy of the result in pinned host memory itself -- a list against itself in K3's
            # streamed form -- the columns are filled from there; otherwise every range is downloaded on a side stream)
            res, h_idx, h_val = _lib.cossim_topn_ranges(ctx, self._dev_index, from_dev, top_n, lower, self_match, ends, _SPLIT_EVENT,
                                                        mirror=True)
            if not h_idx:
                res.rows_begin(0, ends[0], _SPLIT_EVENT, 0)
        else:
            res = _lib.cossim_topn(ctx, self._dev_index, from_dev, max(top_n, 1), lower, exclude_diag=self_match)
        t1 = time.perf_counter()
        pending = ranged and split and bool(h_idx)
        from_col = col[0] if col and col[0] is not None else None if pending else object_column(from_list)        # (host work while the device runs K3)
        t2 = time.perf_counter()
        if split:
            fb = FrameBuilder(from_list, names, top_n, from_col, from_pending=pending)
            waited = framed = 0.0
            tp = t2
            row0 = 0
            trace = [("enqueued", t1 - t0), ("frame wrapped", time.perf_counter() - t0)] if _TRACE else None
            if h_idx and _RANGE_FILL and hasattr(_lib._pack, "fill_ranges"):
                # the whole result arrives in the pinned mirror: the frame's threads wait for the ranges themselves and gather each
                # as soon as it is final (_pack.fill_ranges) -- no Python between a range's announcement and its columns
                wait_addr, ctx_addr = ctx.event_wait_fn()
                stamps = np.zeros(2 * len(ends)) if _TRACE else None
                fb.fill_ranges(h_idx, h_val, ends, wait_addr, ctx_addr, _SPLIT_EVENT, stamps)
                tb = time.perf_counter()
                if trace is not None:
                    off = time.perf_counter() - time.monotonic()       # (the same clock on Linux; the difference is the call's cost)
                    for i in range(len(ends)):
                        trace += [(f"range {i} [{ends[i - 1] if i else 0}, {ends[i]}) here", stamps[2 * i] + off - t0),
                                  (f"range {i} filled", stamps[2 * i + 1] + off - t0)]
                    trace.append(("fill returned", tb - t0))
                framed, tp = framed + (tb - tp), tb
                ends = ()
            for i, row1 in enumerate(ends):
                if h_idx:
                    ctx.event_wait(_SPLIT_EVENT + i)      # (polls the range's word in pinned memory)
                    idx_addr, val_addr = h_idx + 4 * top_n * row0, h_val + 4 * top_n * row0
                else:
                    idx_addr, val_addr = res.rows_finish(i & 1)
                    if i + 1 < len(ends):         # the next range's copies are on their way while this one's columns are filled
                        res.rows_begin(row1, ends[i + 1], _SPLIT_EVENT + i + 1, (i + 1) & 1)
                ta = time.perf_counter()
                fb.fill_raw(idx_addr, val_addr, row1 - row0, row0)
                tb = time.perf_counter()
                if trace is not None:
                    trace += [(f"range {i} [{row0}, {row1}) here", ta - t0), (f"range {i} filled", tb - t0)]
                waited, framed, tp = waited + (ta - tp), framed + (tb - ta), tb
                row0 = row1
            self.last_trace = trace
            frame = fb.frame()
            t4 = time.perf_counter()
            framed += t4 - tp
        else:
            idx, val = res.download()
            t3 = time.perf_counter()
            frame = topn_to_frame(idx, val, from_list, names, top_n, from_col=from_col)
            t4 = time.perf_counter()
            waited, framed = t3 - t2, t4 - t3
        # where the wall time of the last match went (ms): pack + upload + enqueue of K1/K2/index/K3, the From
        # column (overlapped with the device), waiting for the device + D2H, the remaining frame columns
        self.last_timings = {"upload_and_enqueue": (t1 - t0) * 1e3, "from_column": (t2 - t1) * 1e3,
                             "wait_and_download": waited * 1e3, "frame": framed * 1e3}
        return frame

    def match_device(self, from_list: List[str], to_list: List[str] = None, re_train: bool = True):
        """The match of `match()` left on the device: a _lib.DeviceTopN of (index into the to-list, fp32 cosine)
        per from-string and rank -- for consumers that reduce it further there (polyfuzz_amd.linkage.group_top1)."""
        if self.cosine_method not in _METHODS:
            raise ValueError(f"cosine_method must be one of {_METHODS}")
        ctx = _lib.Context.default()
        from_dev, _ = self._extract_tf_idf(from_list, to_list, re_train)
        top_n = clip_top_n(self.top_n, to_list)
        lower = float(self.min_similarity) if self.cosine_method in ("sparse", "hip") else 0.0
        return _lib.cossim_topn(ctx, self._dev_index, from_dev, max(top_n, 1), lower, exclude_diag=to_list is None)

    # ---- internals ---------------------------------------------------------------
    def _params(self):
        lo, hi = int(self.n_gram_range[0]), int(self.n_gram_range[1])
        return _lib.TfidfParams(lo, hi, int(bool(self.clean_string)), int(bool(self.remove_space_ngrams)))

    def _upload(self, strings, objects=None):
        """objects: see _lib.pack_strings (the From column, filled in the packer's walk over the strings)"""
        ctx = _lib.Context.default()
        direct = _lib.DeviceStrings.upload_direct(ctx, strings, objects) if len(strings) >= 1024 and _DIRECT_PACK else None
        if direct is not None:          # (1-byte strings packed straight into the pinned staging buffer)
            return direct
        packed = _lib.pack_strings(strings, objects)
        if self.clean_string and packed[2] != 1:
            # code points > 0xFF: clean on the host (str.lower() can map into ASCII); cleaning is idempotent,
            # so the device's clean pass leaves these untouched
            packed = _lib.pack_strings([_clean_string(s) for s in strings])
        return _lib.DeviceStrings.upload_packed(ctx, *packed)

    def _extract_tf_idf(self, from_list, to_list=None, re_train=True, from_objects=None):
        """ reference _tfidf.py:102-118: fit on to_list + from_list (or from_list alone), keep the
        to-side matrix; with re_train=False reuse the fitted vocabulary and to-side.
        from_objects: a one-element list holding a fresh object array -- the frame's From column, filled while from_list
        is packed for its upload; left None in the list when from_list was not uploaded by this call """
        ctx = _lib.Context.default()
        self._restore()
        objs = from_objects[0] if from_objects else None
        if from_objects:
            from_objects[0] = None
        if to_list:
            from_s = self._upload(from_list, objs)
            if from_objects:
                from_objects[0] = objs
            if re_train:
                to_s = self._upload(to_list)
                self._fit(ctx, to_s, from_s)
                self._set_to_side(ctx, self._dev_vec.transform(to_s))
            self._require_fitted()
            return self._dev_vec.transform(from_s), self._dev_to
        if re_train:
            from_s = self._upload(from_list, objs)
            if from_objects:
                from_objects[0] = objs
            self._fit(ctx, from_s, None)
            self._set_to_side(ctx, self._dev_vec.transform(from_s))
        self._require_fitted()
        return self._dev_to, self._dev_to

    def _fit(self, ctx, docs_a, docs_b):
        self._dev_vec = _lib.DeviceTfidf.fit(ctx, self._params(), docs_a, docs_b)
        self._vectorizer = HipTfidfVectorizer(self, self._dev_vec)
        self._host_state = None

    def _set_to_side(self, ctx, dev_csr):
        self._dev_to = dev_csr
        self._dev_index = _lib.DeviceIndex.build(ctx, dev_csr)

    def _require_fitted(self):
        if self._dev_vec is None or self._dev_index is None:
            raise ValueError("This TFIDF instance is not fitted yet: call match(..., re_train=True) first")

    # ---- persistence (joblib.dump / load, reference polyfuzz.py:429-457) ---------
    def __getstate__(self):
        state = {k: v for k, v in self.__dict__.items()
                 if k not in ("_dev_vec", "_dev_to", "_dev_index", "_vectorizer", "_host_state")}
        host = self._host_state
        if host is None and self._dev_vec is not None:
            ngrams, idf, _ = self._dev_vec.export()
            host = {"ngrams": ngrams, "idf": idf, "n_docs": self._dev_vec.info()["n_docs"],
                    "params": tuple(getattr(self._dev_vec.params, f) for f, _ in _lib.TfidfParams._fields_),
                    "to_csr": None if self._dev_to is None else self._dev_to.download()}
        state["_host_state"] = host
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._dev_vec = self._dev_to = self._dev_index = self._vectorizer = None

    def _restore(self):
        """Re-create the device state of an unpickled matcher on first use."""
        if self._dev_vec is not None or self._host_state is None:
            return
        ctx = _lib.Context.default()
        h = self._host_state
        self._dev_vec = _lib.DeviceTfidf.from_state(ctx, _lib.TfidfParams(*h["params"]), h["ngrams"], h["idf"],
                                                    h["n_docs"])
        self._vectorizer = HipTfidfVectorizer(self, self._dev_vec)
        if h["to_csr"] is not None:
            indptr, indices, data, n_cols = h["to_csr"]
            self._set_to_side(ctx, _lib.DeviceCSR.upload(ctx, indptr, indices, data, n_cols))
