"""EditDistance -- the reference's all-pairs edit-distance matcher
(polyfuzz/models/_distance.py:14-102) on the MI355X engine.

The reference scores every (from, to) pair with `scorer` (default
rapidfuzz.fuzz.ratio, _distance.py:32) in a Python loop / joblib pool and keeps the
first arg-max per from-string.  Here the whole score matrix + arg-max is one HIP
kernel (K4: bit-parallel LCS, fused first-max reduction); only the Indel ratio
(`fuzz.ratio`) is implemented on the device, so any other scorer raises -- there is
no CPU path to fall back to.
"""
from typing import Callable, List, Union

import numpy as np
import pandas as pd

from .. import _lib
from ._base import BaseMatcher


def _is_ratio(scorer) -> bool:
    if scorer is None or scorer == "ratio":
        return True
    return getattr(scorer, "__name__", "") == "ratio" and "rapidfuzz" in (getattr(scorer, "__module__", "") or "")


class EditDistance(BaseMatcher):
    """
    Calculate the Edit Distance between lists of strings (Indel ratio, rapidfuzz.fuzz.ratio)

    Arguments (reference _distance.py:18-23):
        n_jobs: accepted for compatibility; the GPU kernel ignores it
        scorer: "ratio" / rapidfuzz.fuzz.ratio (default).  Other scorers are not implemented on the device.
        model_id: The name of the particular instance, used when comparing models
        normalize: Whether to min-max normalize the similarity scores (_distance.py:83-86)
    """
    def __init__(self,
                 n_jobs: int = 1,
                 scorer: Union[Callable, str, None] = "ratio",
                 model_id: str = None,
                 normalize: bool = True):
        super().__init__(model_id)
        self.type = "EditDistance"
        if not _is_ratio(scorer):
            raise NotImplementedError(
                "polyfuzz_amd.EditDistance runs rapidfuzz.fuzz.ratio (Indel ratio) on the GPU; "
                f"scorer {scorer!r} has no HIP kernel and there is no CPU fallback")
        self.scorer = scorer
        self.normalize = normalize
        self.n_jobs = n_jobs

    def match(self,
              from_list: List[str],
              to_list: List[str] = None,
              **kwargs) -> pd.DataFrame:
        """ Best match (first maximum of the ratio) of every from-string (reference _distance.py:46-87) """
        idx, score, names = self._best(from_list, to_list)
        to_col = [names[j] for j in idx.tolist()]
        matches = pd.DataFrame({"From": pd.Series(list(from_list), dtype=object),
                                "To": pd.Series(to_col, dtype=object),
                                "Similarity": score})
        if self.normalize:      # global min-max over the best scores, _distance.py:83-86
            matches["Similarity"] = (matches["Similarity"] -
                                     matches["Similarity"].min()) / (matches["Similarity"].max() -
                                                                     matches["Similarity"].min())
        return matches

    def _best(self, from_list, to_list, rows=None):
        ctx = _lib.Context.default()
        self_match = to_list is None
        names = list(from_list) if self_match else list(to_list)
        skip = None
        if self_match:
            # list.remove(from_string) drops the FIRST equal element (_distance.py:93-96)
            first = {}
            for j, s in enumerate(names):
                first.setdefault(s, j)
            skip = np.fromiter((first[s] for s in from_list), np.int32, len(from_list))
        if len(names) - (1 if self_match else 0) <= 0 and len(from_list) > 0:
            raise ValueError("attempt to get argmax of an empty sequence")   # np.argmax([]) in the reference
        f_dev = _lib.DeviceStrings.upload(ctx, list(from_list))
        t_dev = f_dev if self_match else _lib.DeviceStrings.upload(ctx, names)
        begin, end = (0, len(from_list)) if rows is None else rows
        idx, score = _lib.indel_argmax(ctx, f_dev, t_dev, skip, begin, end)
        return idx, score, names
