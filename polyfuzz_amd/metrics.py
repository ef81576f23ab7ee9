"""precision_recall_curve -- the reference's threshold sweep over the Similarity column
(polyfuzz/metrics.py:12-53) on the MI355X engine.

Same signature and return value.  The reference loops over the 101 thresholds and
re-filters the whole column for each; here one HIP kernel (K6, csrc/k6_reductions.hip)
bins every similarity by the number of thresholds it passes and the suffix sums of
the bins give "count >= p" and "sum of those" for all thresholds at once.
(`visualize_precision_recall` -- matplotlib/seaborn plots -- is outside the hot path.)
"""
from typing import List, Tuple

import numpy as np
import pandas as pd

from . import _lib


def precision_recall_curve(matches: pd.DataFrame,
                           precision_steps: float = 0.01) -> Tuple[List[float], List[float], List[float]]:
    """ Calculate precision recall curve based on minimum similarity between strings

    Arguments (reference metrics.py:12-37):
        matches: contains the columns *From*, *To*, and *Similarity*
        precision_steps: the incremental steps in minimum precision

    Returns:
        min_precisions: minimum precision steps
        recall: recall per minimum precision step (share of matches with Similarity >= step)
        average_precision: mean Similarity of those matches (nan when there are none)
    """
    min_precisions = list(np.arange(0., 1 + precision_steps, precision_steps))
    similarities = np.asarray(matches.Similarity.values, np.float64)
    total = len(matches)
    count, ssum = _lib.pr_curve(_lib.Context.default(), similarities, np.asarray(min_precisions, np.float64))
    recall = [int(c) / total for c in count]                     # ZeroDivisionError on an empty frame, as the reference
    average_precision = [float(s / c) if c else float("nan") for s, c in zip(ssum.tolist(), count.tolist())]
    return min_precisions, recall, average_precision
