"""RapidFuzz -- the reference's `process.extractOne` matcher (polyfuzz/models/_rapidfuzz.py:10-113),
what `PolyFuzz("EditDistance")` dispatches to (polyfuzz.py:128-130), on the MI355X engine.

Same constructor and `.match()` contract: per from-string the best choice of the to-list under `scorer`
(first maximum, as rapidfuzz.process.extractOne keeps the first best), `None` / 0.0 when the best score is
below `score_cutoff`, Similarity = score / 100.

Scorers, all on the device (names or the rapidfuzz.fuzz functions of those names):
* `ratio`, `QRatio` (= ratio, but 0 when either string is empty), `token_sort_ratio` (= ratio of the strings'
  whitespace tokens sorted and joined by one space) -- one fixed string per list element: K4 (k4_indel.hip);
* `WRatio` (the reference's DEFAULT), `partial_ratio`, `token_set_ratio`, `token_ratio`, `partial_token_sort_ratio`,
  `partial_token_set_ratio`, `partial_token_ratio` -- a different string pair for every (from, to): K7
  (k7_fuzz.hip: windows by masked match tables and prefix bit counts, token-set differences by token masks).
rapidfuzz 3.x semantics: no default processor (strings are scored as given).  Any other callable raises
`NotImplementedError` (there is no CPU path in this package).
Strings of any length and token count are accepted; from-strings beyond 256 characters or 32 distinct tokens (and
to-strings beyond 32 distinct tokens) take K7's general kernel, which is slow.

Self-match.  The reference removes the from-string from ONE shared copy of the list (`to_list.remove(from_string)`,
_rapidfuzz.py:103-104), so with n_jobs=1 the list shrinks as rows are processed: row i is scored against the strings AFTER
it only, the last row against nothing.  By default a self-match here excludes only the first list element equal to the
from-string, for that from-string (the behaviour of EditDistance, _distance.py:93-96; SURVEY App. B calls the shrinking list
a bug).  `reference_self_match = True` (attribute, or keyword of `match`) reproduces the reference's frame instead -- on the
device: every choice up to and including the row itself is left out (skip codes, csrc/pfz_internal.h) -- held to frames of
the reference class itself (tests/golden/make_golden_rapidfuzz_self.py).

PARITY UNPINNED: rapidfuzz is not installable in the build container; the scorer is the oracle's restatement
(oracle/indel.c), extractOne's tie / cut-off rules are restated from rapidfuzz's documentation.
"""
from typing import Callable, List, Union

import numpy as np
import pandas as pd

from .. import _lib
from ._base import BaseMatcher
from ._utils import pair_frame, pair_frame_blocks

_K4_SCORERS = ("ratio", "QRatio", "token_sort_ratio")
_DEVICE_SCORERS = _K4_SCORERS + tuple(_lib.FUZZ_SCORERS)


def _scorer_name(scorer) -> str:
    """Name of the rapidfuzz.fuzz scorer `scorer` stands for; a callable from anywhere else (Levenshtein.ratio on
    its 0..1 scale, a user's own `ratio`) must not be mistaken for the rapidfuzz function of the same name."""
    if scorer is None:
        return "WRatio"                                   # the reference's default, fuzz.WRatio
    if isinstance(scorer, str):
        return scorer
    if "rapidfuzz" not in (getattr(scorer, "__module__", "") or ""):
        return repr(scorer)
    return getattr(scorer, "__name__", repr(scorer))


def _left_out(choice: int, skip: int) -> bool:
    """the skip codes of the best-choice kernels (csrc/pfz_internal.h: choice_left_out)"""
    return choice == skip or choice <= -2 - skip


def _transform(name, strings):
    """the string every list element is scored as: itself, or (token_sort_ratio) its whitespace tokens sorted and joined"""
    if name == "token_sort_ratio":           # rapidfuzz: ratio(" ".join(sorted(s1.split())), " ".join(sorted(s2.split())))
        return [" ".join(sorted(s.split())) for s in strings]
    return strings


def upload_for(ctx, name, strings):
    """A list resident on the device in the form scorer `name` reads (K4 scorers: one fixed string per element; K7 scorers:
    the strings themselves -- their token forms are built on the device and cached on the handle)."""
    return _lib.DeviceStrings.upload(ctx, _transform(name, strings))


class _PendingBest:
    """an enqueued best-choice pass: `result()` waits for the device and returns (index int32[n], score float64[n])"""

    def __init__(self, out, n, fix=None):
        self.out, self.n, self.fix = out, n, fix

    def result(self):
        idx, score = _lib.best_from_topn(*self.out.download())
        idx, score = idx[:self.n], score[:self.n]
        if self.fix is not None:
            self.fix(idx, score)
        return idx, score


def best_choice_async(ctx, name, from_list, names, skip, self_match, to_dev=None):
    """Enqueue the best-choice pass of every from-string under the rapidfuzz.fuzz scorer `name` and return at once (the
    caller builds its From column while the device works); `names` are the choices (the from-list itself in a
    self-match, where skip[i] is the choice left out for from-string i -- still the from-string's own first occurrence
    in the ORIGINAL list).  to_dev: the choices already resident (upload_for), e.g. from the previous call of a fitted
    matcher."""
    n = len(from_list)
    f_dev = upload_for(ctx, name, from_list)
    t_dev = f_dev if self_match else (to_dev if to_dev is not None else upload_for(ctx, name, names))
    out = _lib.DeviceTopN.alloc(ctx, max(n, 1), 2)
    fix = None
    if name in _lib.FUZZ_SCORERS:
        _lib.fuzz_extract_one_dev(ctx, f_dev, t_dev, name, out, skip)
    else:
        _lib.indel_argmax_dev(ctx, f_dev, t_dev, out, skip)
        if name == "QRatio":
            def fix(idx, score):
                # QRatio differs from ratio only when BOTH strings are empty (0 instead of 100): an empty from-string
                # scores 0 against every choice, so its first best is simply its first choice
                for i in [i for i, s in enumerate(from_list) if len(s) == 0]:
                    first_choice = next((j for j in range(len(names)) if not (skip is not None and _left_out(j, int(skip[i])))), -1)
                    idx[i], score[i] = first_choice, 0.0
    return _PendingBest(out, n, fix)


def best_choice(ctx, name, from_list, names, skip, self_match, to_dev=None):
    """(index of the first best choice int32[n], its score float64[n] on the 0..100 scale) of every from-string"""
    return best_choice_async(ctx, name, from_list, names, skip, self_match, to_dev).result()


class RapidFuzz(BaseMatcher):
    """
    Calculate the Edit Distance between lists of strings using RapidFuzz's process function

    Arguments (reference _rapidfuzz.py:17-38):
        n_jobs: accepted for compatibility; the GPU kernel ignores it
        score_cutoff: The minimum similarity for which to return a good match. Should be between 0 and 1.
        scorer: a rapidfuzz.fuzz scorer or its name; default (None) = fuzz.WRatio as in the reference.  Every
                rapidfuzz.fuzz scorer runs on the device; other callables raise NotImplementedError
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
                f"polyfuzz_amd.RapidFuzz computes the rapidfuzz.fuzz scorers {_DEVICE_SCORERS} on the GPU; there is no "
                f"CPU path for an arbitrary scorer such as {name!r} -- use the reference's RapidFuzz matcher for it")
        self.scorer = scorer
        self._scorer_name = name
        self.n_jobs = n_jobs
        self.reference_self_match = False        # True: a self-match as the reference's shared, shrinking list gives it (n_jobs = 1)
        self._to_dev = self._to_names = None     # device copy (+ cached plan) of the last to-list

    def match(self,
              from_list: List[str],
              to_list: List[str] = None,
              **kwargs) -> pd.DataFrame:
        """ Best choice of the to-list for every from-string (reference _rapidfuzz.py:61-113).

        The reference scores against the to_list it is handed, always (`PolyFuzz.transform` passes `self.to_list`,
        polyfuzz.py:234-240).  `re_train=False` only lets this matcher re-use what is resident: when the list it is handed IS
        the previous call's (the same object, or an equal list), its device copy, token forms and plan are used again -- no
        upload, no preparation.  Any other list is uploaded. """
        ctx = _lib.Context.default()
        self_match = to_list is None
        # the resident copy stands for the list it was made from and for no other
        # -- compared by CONTENT against a snapshot taken when it was uploaded (the caller's list may have been changed
        # in place since; an ndarray / Series to-list has no list `==`)
        snap = None if self_match else tuple(to_list)
        reuse = kwargs.get("re_train", True) is False and not self_match and self._to_dev is not None and snap == self._to_names
        held = self._to_dev
        self._to_dev = self._to_names = None      # set again below, once this call's to-list is resident
        names = from_list if self_match else to_list
        n = len(from_list)
        skip = None
        if self_match and kwargs.get("reference_self_match", self.reference_self_match):
            # _rapidfuzz.py:86-104 with n_jobs = 1: when row i is scored, list.remove has taken the rows 0 .. i out of the shared
            # list (the first remaining element equal to from_list[i] is row i itself), and what is left keeps its order
            skip = (-2 - np.arange(n, dtype=np.int64)).astype(np.int32)
        elif self_match:
            first = {}
            for j, s in enumerate(names):
                first.setdefault(s, j)
            skip = np.fromiter((first[s] for s in from_list), np.int32, n)
        blocks = None
        if n == 0 or len(names) - (1 if self_match else 0) <= 0:
            idx, score = np.full(n, -1, np.int32), np.zeros(n)             # extractOne over no choices: None
            blocks = pair_frame_blocks(from_list)
        else:
            if not self_match:
                self._to_dev, self._to_names = (held if reuse else upload_for(ctx, self._scorer_name, names)), snap
            pending = best_choice_async(ctx, self._scorer_name, from_list, names, skip, self_match,
                                        to_dev=None if self_match else self._to_dev)
            blocks = pair_frame_blocks(from_list)                            # (the From column: host work while the device scores)
            idx, score = pending.result()
        hit = (idx >= 0) & (score >= self.score_cutoff)                       # extractOne: best score >= score_cutoff
        sim = np.where(hit, score / 100, 0.0)
        return pair_frame(from_list, names, idx, sim, keep=hit, blocks=blocks)

    # a matcher is pickled by joblib (reference polyfuzz.py:429-457): device handles stay behind
    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in ("_to_dev", "_to_names")}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.setdefault("reference_self_match", False)      # (a matcher pickled before the attribute existed)
        self._to_dev = self._to_names = None
