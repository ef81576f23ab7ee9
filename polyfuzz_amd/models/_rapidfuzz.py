"""RapidFuzz -- the reference's `process.extractOne` matcher (polyfuzz/models/_rapidfuzz.py:10-113),
what `PolyFuzz("EditDistance")` dispatches to (polyfuzz.py:128-130), on the MI355X engine.

Same constructor and `.match()` contract: per from-string the best choice of the to-list under `scorer`
(first maximum, as rapidfuzz.process.extractOne keeps the first best), `None` / 0.0 when the best score is
below `score_cutoff`, Similarity = score / 100.

Scorers on the device: the Indel-ratio family K4 computes -- `fuzz.ratio` and `fuzz.QRatio` (= ratio, but 0 when
either string is empty).  The reference's DEFAULT, `fuzz.WRatio`, and the partial_* / token_* scorers build a
different string pair for every (from, to) -- token-set differences, sliding windows -- which the
from-string-stationary bit-parallel kernel cannot express; they raise `NotImplementedError` (there is no CPU
path in this package; run the reference's own RapidFuzz matcher for those).

Deviation, on purpose: the reference removes the from-string from ONE shared copy of the list
(`to_list.remove(from_string)`, _rapidfuzz.py:103-104), so with n_jobs=1 the list shrinks as rows are processed
and later from-strings can no longer match earlier ones.  Here a self-match excludes only the first list element
equal to the from-string, for that from-string (the behaviour of EditDistance, _distance.py:93-96).

PARITY UNPINNED: rapidfuzz is not installable in the build container; the scorer is the oracle's restatement
(oracle/indel.c), extractOne's tie / cut-off rules are restated from rapidfuzz's documentation.
"""
from typing import Callable, List, Union

import numpy as np
import pandas as pd

from .. import _lib
from ._base import BaseMatcher
from ._utils import object_column

_DEVICE_SCORERS = ("ratio", "QRatio")


def _scorer_name(scorer) -> str:
    if scorer is None:
        return "WRatio"                                   # the reference's default, fuzz.WRatio
    if isinstance(scorer, str):
        return scorer
    return getattr(scorer, "__name__", repr(scorer))


class RapidFuzz(BaseMatcher):
    """
    Calculate the Edit Distance between lists of strings using RapidFuzz's process function

    Arguments (reference _rapidfuzz.py:17-38):
        n_jobs: accepted for compatibility; the GPU kernel ignores it
        score_cutoff: The minimum similarity for which to return a good match. Should be between 0 and 1.
        scorer: "ratio" / fuzz.ratio or "QRatio" / fuzz.QRatio run on the device; the reference's default
                fuzz.WRatio (scorer=None here) and the partial / token scorers raise NotImplementedError
        model_id: The name of the particular instance, used when comparing models
    """
    def __init__(self,
                 n_jobs: int = 1,
                 score_cutoff: float = 0,
                 scorer: Union[Callable, str, None] = None,
                 model_id: str = None):
        super().__init__(model_id)
        self.type = "EditDistance"
        self.score_cutoff = score_cutoff * 100
        name = _scorer_name(scorer)
        if name not in _DEVICE_SCORERS:
            raise NotImplementedError(
                f"polyfuzz_amd.RapidFuzz computes {_DEVICE_SCORERS} (Indel ratio) on the GPU; scorer {name!r} "
                "(WRatio is the reference's default) builds per-pair strings the kernel cannot express and there is no CPU "
                "fallback -- pass scorer='ratio', or use the reference's RapidFuzz matcher")
        self.scorer = scorer
        self._scorer_name = name
        self.n_jobs = n_jobs

    def match(self,
              from_list: List[str],
              to_list: List[str] = None,
              **kwargs) -> pd.DataFrame:
        """ Best choice of the to-list for every from-string (reference _rapidfuzz.py:61-113) """
        ctx = _lib.Context.default()
        self_match = to_list is None
        names = from_list if self_match else to_list
        n = len(from_list)
        skip = None
        if self_match:
            first = {}
            for j, s in enumerate(names):
                first.setdefault(s, j)
            skip = np.fromiter((first[s] for s in from_list), np.int32, n)
        if n == 0 or len(names) - (1 if self_match else 0) <= 0:
            idx, score = np.full(n, -1, np.int32), np.zeros(n)             # extractOne over no choices: None
        else:
            f_dev = _lib.DeviceStrings.upload(ctx, from_list)
            t_dev = f_dev if self_match else _lib.DeviceStrings.upload(ctx, names)
            idx, score = _lib.indel_argmax(ctx, f_dev, t_dev, skip)
        if self._scorer_name == "QRatio":
            # QRatio differs from ratio only when BOTH strings are empty (0 instead of 100): an empty from-string
            # scores 0 against every choice, so its first best is simply its first choice
            for i in [i for i, s in enumerate(from_list) if len(s) == 0]:
                first_choice = next((j for j in range(len(names)) if not (self_match and j == skip[i])), -1)
                idx[i], score[i] = first_choice, 0.0
        hit = (idx >= 0) & (score >= self.score_cutoff)                       # extractOne: best score >= score_cutoff
        to_col = object_column([names[j] if ok else None for j, ok in zip(idx.tolist(), hit.tolist())])
        sim = np.where(hit, score / 100, 0.0)
        return pd.DataFrame({"From": object_column(from_list), "To": to_col, "Similarity": sim}, copy=False)
