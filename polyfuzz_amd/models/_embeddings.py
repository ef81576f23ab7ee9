"""Embeddings -- the reference's embedding matcher (polyfuzz/models/_embeddings.py:15-145) on the MI355X
engine, for the part of it that is on the hot path: cosine top-n over ready-made embedding matrices
(SURVEY.md §8 row A12).

The reference embeds the strings with Flair/word-embedding models that have to be downloaded; those
models are out of scope.  What is mirrored here is the matcher: `match(from_list, to_list,
embeddings_from=..., embeddings_to=..., re_train=...)` with the reference's argument meaning, the stored
`embeddings_to` for `re_train=False`, and the similarity operator on dense input.  An `embedding_method`
may be any callable `list[str] -> ndarray[n, d]` (the rows are L2-normalised like _embeddings.py:145);
without one, the embeddings must be passed in.
"""
from typing import Callable, List, Optional

import numpy as np
import pandas as pd

from .. import _lib
from ._base import BaseMatcher
from ._utils import topn_to_frame, clip_top_n, _METHODS


class Embeddings(BaseMatcher):
    """
    Match two lists of strings by the cosine similarity of their embeddings

    Arguments (reference _embeddings.py:60-65):
        embedding_method: callable mapping a list of strings to an ndarray of row vectors, or None when
                          the embeddings are always supplied to `match`
        min_similarity: The minimum similarity between strings, otherwise return 0 similarity
        top_n: The number of best matches you want returned
        cosine_method: "sparse" (raw dot product of the vectors as given, honours `min_similarity`),
                       "sklearn" / "knn" (true cosine, ignore `min_similarity`) -- the reference's
                       semantics, _utils.py:59-102 -- or "hip" (true cosine, honours `min_similarity`)
        model_id: The name of the particular instance, used when comparing models
    """
    def __init__(self,
                 embedding_method: Optional[Callable[[List[str]], np.ndarray]] = None,
                 min_similarity: float = 0.75,
                 top_n: int = 1,
                 cosine_method: str = "sparse",
                 model_id: str = None):
        super().__init__(model_id)
        self.type = "Embeddings"
        if embedding_method is not None and not callable(embedding_method):
            raise TypeError("polyfuzz_amd.Embeddings: embedding_method must be a callable list[str] -> ndarray "
                            "(Flair embedding objects are not supported: their models need downloads)")
        self.embedding_method = embedding_method
        self.min_similarity = min_similarity
        self.top_n = top_n
        self.cosine_method = cosine_method
        self.embeddings_to = None
        self._dev_to = None            # _lib.DeviceDense of the to-side
        self._dev_to_normalize = None

    def match(self,
              from_list: List[str],
              to_list: List[str] = None,
              embeddings_from: np.ndarray = None,
              embeddings_to: np.ndarray = None,
              re_train: bool = True) -> pd.DataFrame:
        """ Matches the two lists of strings to each other and returns the best mapping
        (reference _embeddings.py:87-135) """
        if not isinstance(embeddings_from, np.ndarray):
            embeddings_from = self._embed(from_list)
        explicit_to = isinstance(embeddings_to, np.ndarray)     # the caller's own to-side always wins (_embeddings.py:117-133)
        if not explicit_to:
            if not re_train:
                embeddings_to = self.embeddings_to
                if embeddings_to is None:
                    raise ValueError("This Embeddings instance holds no to-side embeddings yet: "
                                     "call match(..., re_train=True) first")
            elif to_list is None:
                embeddings_to = self._embed(from_list) if self.embedding_method is not None else embeddings_from
            else:
                embeddings_to = self._embed(to_list)
        if self.cosine_method not in _METHODS:
            raise ValueError(f"cosine_method must be one of {_METHODS}")
        ctx = _lib.Context.default()
        normalize = self.cosine_method != "sparse"        # "sparse": raw dot products (reference _utils.py:74-82)
        lower = float(self.min_similarity) if self.cosine_method in ("sparse", "hip") else 0.0
        # the to-side stays in HBM: match(..., re_train=False) (PolyFuzz.transform, polyfuzz.py:234-240) uploads
        # the new from-vectors only; an explicitly passed to-side is uploaded unless it IS the resident one
        stale = explicit_to and embeddings_to is not self.embeddings_to
        if re_train or stale or self._dev_to is None or self._dev_to_normalize != normalize:
            self._dev_to = _lib.DeviceDense.upload(ctx, np.asarray(embeddings_to), normalize)
            self._dev_to_normalize = normalize
        self_match = to_list is None
        same = self_match and embeddings_to is embeddings_from
        from_dev = self._dev_to if same else _lib.DeviceDense.upload(ctx, np.asarray(embeddings_from), normalize)
        if from_dev.dim != self._dev_to.dim:
            raise ValueError(f"dense cosine needs two 2-D arrays with equal width, got {from_dev.dim} and {self._dev_to.dim}")
        top_n = clip_top_n(self.top_n, to_list)
        idx, val = _lib.dense_topn(ctx, from_dev, self._dev_to, max(top_n, 1), lower, exclude_diag=self_match).download()
        self.embeddings_to = embeddings_to
        return topn_to_frame(idx, val, from_list, from_list if self_match else to_list, top_n)

    # a matcher is pickled by joblib (reference polyfuzz.py:429-457): the device copy stays behind
    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in ("_dev_to",)}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._dev_to = None

    def _embed(self, strings: List[str]) -> np.ndarray:
        """ Embed with the user's callable and L2-normalise the rows (reference _embeddings.py:136-145) """
        if self.embedding_method is None:
            raise ValueError("polyfuzz_amd.Embeddings was created without an embedding_method: pass "
                             "embeddings_from / embeddings_to to match()")
        vec = np.asarray(self.embedding_method(list(strings)), dtype=np.float64)
        if vec.ndim != 2 or vec.shape[0] != len(strings):
            raise ValueError(f"embedding_method returned shape {vec.shape} for {len(strings)} strings")
        norms = np.sqrt((vec * vec).sum(axis=1, keepdims=True))
        norms[norms == 0.0] = 1.0
        return vec / norms
