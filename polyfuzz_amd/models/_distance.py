"""EditDistance -- the reference's all-pairs edit-distance matcher
(polyfuzz/models/_distance.py:14-102) on the MI355X engine.

The reference scores every (from, to) pair with `scorer` (default
rapidfuzz.fuzz.ratio, _distance.py:32) in a Python loop / joblib pool and keeps the
first arg-max per from-string.  Here the whole score matrix + arg-max is one HIP
kernel (K4: bit-parallel LCS, fused first-max reduction).  `scorer` may be any
rapidfuzz.fuzz scorer (or its name): ratio / QRatio / token_sort_ratio run through K4,
WRatio / partial_ratio / token_set_ratio / token_ratio / partial_token_* through K7;
any other callable raises -- there is no CPU path to fall back to.
"""
import time
from typing import Callable, List, Union

import numpy as np
import pandas as pd

from .. import _lib
from ._base import BaseMatcher
from ._utils import pair_frame, pair_frame_blocks


def _device_scorer(scorer) -> str:
    """Name of the rapidfuzz.fuzz scorer `scorer` stands for, or '' when it has no kernel."""
    from ._rapidfuzz import _DEVICE_SCORERS
    if scorer is None:
        return "ratio"                                     # the reference's default (_distance.py:32)
    name = scorer if isinstance(scorer, str) else getattr(scorer, "__name__", "")
    if not isinstance(scorer, str) and "rapidfuzz" not in (getattr(scorer, "__module__", "") or ""):
        return ""
    return name if name in _DEVICE_SCORERS else ""


class EditDistance(BaseMatcher):
    """
    Calculate the Edit Distance between lists of strings (Indel ratio, rapidfuzz.fuzz.ratio)

    Arguments (reference _distance.py:18-23):
        n_jobs: accepted for compatibility; the GPU kernel ignores it
        scorer: a rapidfuzz.fuzz scorer or its name; default "ratio" / rapidfuzz.fuzz.ratio.  Other callables
                have no kernel (NotImplementedError).
        model_id: The name of the particular instance, used when comparing models
        normalize: Whether to min-max normalize the similarity scores (_distance.py:83-86)

    scorer "ratio" (K4): from-strings of up to 1024 characters run in the register-resident word classes (eight / four at
    a time up to 16 / 32 characters); longer ones, and alphabets whose match table (distinct code points of the to-list x
    words) exceeds 60 KiB of LDS, take a general -- slower -- kernel with its state in global memory.  The per-pair
    scorers (K7: WRatio, partial_ratio, the token_set family): from-strings of up to 256 characters and 32 distinct tokens
    in the LDS kernel, anything beyond (and to-strings beyond 32 distinct tokens) in K7's general kernel, which is slow.
    Any length and any alphabet is accepted.  There is no CPU fallback.
    """
    def __init__(self,
                 n_jobs: int = 1,
                 scorer: Union[Callable, str, None] = "ratio",
                 model_id: str = None,
                 normalize: bool = True):
        super().__init__(model_id)
        self.type = "EditDistance"
        self._scorer_name = _device_scorer(scorer)
        if not self._scorer_name:
            raise NotImplementedError(
                "polyfuzz_amd.EditDistance runs the rapidfuzz.fuzz scorers on the GPU; "
                f"scorer {scorer!r} has no HIP kernel and there is no CPU fallback")
        self.scorer = scorer
        self.normalize = normalize
        self.n_jobs = n_jobs
        self._to_dev = self._to_names = None     # device copy (+ cached K4 plan) of the last to-list
        self.last_timings = None

    def match(self,
              from_list: List[str],
              to_list: List[str] = None,
              **kwargs) -> pd.DataFrame:
        """ Best match (first maximum of the ratio) of every from-string (reference _distance.py:46-87).

        The reference scores against the to_list it is handed, always (`PolyFuzz.transform` passes `self.to_list`,
        polyfuzz.py:234-240).  `re_train=False` only lets this matcher re-use what is resident: when the list it is handed
        IS the previous call's (the same object, or an equal list), its device copy and K4 plan (alphabet, length-sorted
        packed groups) are used again -- no upload, no preparation.  Any other list is uploaded. """
        t0 = time.perf_counter()
        pending, names = self._best(from_list, to_list, reuse_to=kwargs.get("re_train", True) is False)
        blocks = pair_frame_blocks(from_list)        # (the From column: host work while the device scores)
        idx, score = pending.result()
        t1 = time.perf_counter()
        matches = pair_frame(from_list, names, idx, score, blocks=blocks)
        if self.normalize:      # global min-max over the best scores, _distance.py:83-86
            matches["Similarity"] = (matches["Similarity"] -
                                     matches["Similarity"].min()) / (matches["Similarity"].max() -
                                                                     matches["Similarity"].min())
        self.last_timings = {"device": (t1 - t0) * 1e3, "frame": (time.perf_counter() - t1) * 1e3}
        return matches

    def _best(self, from_list, to_list, reuse_to=False):
        ctx = _lib.Context.default()
        self_match = to_list is None
        skip = None
        if self_match:
            names = from_list
            # list.remove(from_string) drops the FIRST equal element (_distance.py:93-96)
            first = {}
            for j, s in enumerate(names):
                first.setdefault(s, j)
            skip = np.fromiter((first[s] for s in from_list), np.int32, len(from_list))
        else:
            names = to_list
        # the resident copy stands for the list it was made from and for no other
        # -- compared by CONTENT against a snapshot taken when it was uploaded (the caller's list may have been changed
        # in place since; an ndarray / Series to-list has no list `==`)
        snap = None if self_match else tuple(to_list)
        reuse_to = reuse_to and not self_match and self._to_dev is not None and snap == self._to_names
        held = (self._to_dev, self._to_names)
        self._to_dev = self._to_names = None      # set again below, once this call's to-list is resident
        if len(names) - (1 if self_match else 0) <= 0 and len(from_list) > 0:
            raise ValueError("attempt to get argmax of an empty sequence")   # np.argmax([]) in the reference
        from ._rapidfuzz import best_choice_async, upload_for
        name = self._scorer_name              # "ratio": K4; the other rapidfuzz.fuzz scorers: K4 on transformed strings, or K7
        to_dev = None
        if not self_match:
            if reuse_to:
                to_dev = held[0]
            else:
                to_dev = upload_for(ctx, name, names)
            self._to_dev, self._to_names = to_dev, snap
        return best_choice_async(ctx, name, from_list, names, skip, self_match, to_dev=to_dev), names

    # a matcher is pickled by joblib (reference _distance.py:77, polyfuzz.py:429-457): device handles stay behind
    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in ("_to_dev", "_to_names")}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._to_dev = self._to_names = None
