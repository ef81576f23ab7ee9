"""BaseMatcher -- the plugin ABC of the reference (polyfuzz/models/_base.py:6-31).

When the reference package is importable our matchers subclass ITS BaseMatcher, so
`PolyFuzz(polyfuzz_amd.models.TFIDF(...))` passes the facade's isinstance check
(polyfuzz/polyfuzz.py:141); otherwise an identical ABC is defined here.
"""
from abc import ABC, abstractmethod
from typing import List

import pandas as pd

try:  # pragma: no cover - depends on the environment
    from polyfuzz.models._base import BaseMatcher as _RefBaseMatcher
except Exception:  # reference not installed (or its optional deps missing)
    _RefBaseMatcher = None

if _RefBaseMatcher is not None:
    BaseMatcher = _RefBaseMatcher
else:
    class BaseMatcher(ABC):
        """ The abstract BaseMatching to be modelled after for string matching """

        def __init__(self, model_id: str = "Model 0"):
            self.model_id = model_id
            self.type = "Base Model"

        @abstractmethod
        def match(self, from_list: List[str], to_list: List[str] = None, **kwargs) -> pd.DataFrame:
            """ Arguments: from_list, to_list.  Returns a DataFrame with columns From, To, Similarity. """
            raise NotImplementedError()
