"""single_linkage -- the reference's grouping of matched strings (polyfuzz/linkage.py:5-53),
the consumer of the self-match hot path in `PolyFuzz.group` (polyfuzz.py:331-373,459-484).

Same signature and return values.  The reference walks the filtered frame row by row with a
dict of strings; the assignment is greedy and ORDER-DEPENDENT (a From adopts the cluster of
its To if that exists, else both found a new cluster), and cluster id 0 is falsy in
`if not cluster_mapping.get(...)`, so members of the first cluster are treated as unmapped
whenever they are met again -- the reference's tests pin the resulting ids
(tests/test_polyfuzz.py:85-86: {1: ['apples', 'apple']}).  Union-find / connected components is
NOT equivalent.  Here:

* `single_linkage(matches, min_similarity)`: any frame.  Strings become integer ids (pandas
  factorize), the greedy pass runs on the id arrays -- in the CPython helper `_pack.linkage`, or
  its pure-Python twin -- and the three dicts are rebuilt in the reference's insertion order.
* `group_top1(...)`: the same result for a self-match top-1 device result without building the
  frame first (K6 linkage kernel, csrc/k6_reductions.hip), used by TFIDF-based grouping.
"""
from typing import List, Mapping, Tuple

import numpy as np
import pandas as pd

from . import _lib


def _greedy_py(from_ids, to_ids, n_strings):
    """Pure-Python twin of _pack.linkage (reference linkage.py:28-45 on ids)."""
    cluster = np.full(n_strings, -1, np.int32)
    order = []
    nxt = 0
    for a, b in zip(from_ids.tolist(), to_ids.tolist()):
        if cluster[a] > 0:
            continue
        if cluster[b] <= 0:
            if cluster[b] < 0:
                order.append(b)
            cluster[b] = nxt
            if cluster[a] < 0:
                order.append(a)
            cluster[a] = nxt
            nxt += 1
        else:
            if cluster[a] < 0:
                order.append(a)
            cluster[a] = cluster[b]
    return cluster, np.asarray(order, np.int32)


def greedy_assign(from_ids, to_ids, n_strings):
    """(cluster int32[n_strings] (-1 = never mapped), order int32[k]) -- see _pack.linkage."""
    from_ids = np.ascontiguousarray(from_ids, np.int32)
    to_ids = np.ascontiguousarray(to_ids, np.int32)
    if _lib._pack is None:
        return _greedy_py(from_ids, to_ids, n_strings)
    cl, od = _lib._pack.linkage(from_ids.tobytes(), to_ids.tobytes(), int(n_strings))
    return np.frombuffer(cl, np.int32), np.frombuffer(od, np.int32)


def dicts_from_assignment(strings, cluster, order):
    """The reference's three return values from (cluster id per string id, insertion order)."""
    keys = [strings[i] for i in order.tolist()]
    vals = cluster[order].tolist()
    cluster_mapping = dict(zip(keys, vals))
    clusters = {}
    for key, value in zip(keys, vals):                   # reference linkage.py:47-51
        clusters.setdefault(value, []).append(key)
    cluster_name_map = {key: clusters[value][0] for key, value in zip(keys, vals)}
    return clusters, cluster_mapping, cluster_name_map


def single_linkage(matches: pd.DataFrame,
                   min_similarity: float = 0.8) -> Tuple[Mapping[int, List[str]], Mapping[str, int], Mapping[str, str]]:
    """ Single linkage clustering from column 'From' to column 'To'

    Arguments (reference linkage.py:5-26):
        matches: contains the columns *From*, *To*, and *Similarity* used for creating groups
        min_similarity: minimum similarity between strings before they can be merged into a group

    Returns:
        clusters: The populated clusters
        cluster_mapping: The mapping from a string to a cluster
        cluster_name_map: The mapping from a string to the representative string in its respective cluster
    """
    kept = matches.loc[matches.Similarity > min_similarity, :]
    frm = kept["From"].to_numpy(dtype=object)
    to = kept["To"].to_numpy(dtype=object)
    m = len(kept)
    # one id per distinct value; None (a To without a match that passed a negative threshold) is a key too
    codes, uniques = pd.factorize(np.concatenate([frm, to]), use_na_sentinel=True)
    strings = list(uniques)
    if (codes < 0).any():
        codes = np.where(codes < 0, len(strings), codes)
        strings.append(None)
    cluster, order = greedy_assign(codes[:m], codes[m:], len(strings))
    return dicts_from_assignment(strings, cluster, order)


def group_top1(result, strings: List[str], min_similarity: float):
    """The three dicts of `single_linkage(model.match(strings), min_similarity)` straight from the device-resident
    self-match result of `strings` (a _lib.DeviceTopN; only rank 0 is read): K6's parallel re-statement of the
    greedy walk, no frame in between.  `strings` must be unique (as in PolyFuzz._create_groups, polyfuzz.py:468-471)
    and min_similarity >= 0; otherwise build the frame and call single_linkage."""
    cluster, key, _ = _lib.linkage_top1(result.ctx, result, min_similarity)
    mapped = np.nonzero(cluster >= 0)[0]
    order = mapped[np.argsort(key[mapped], kind="stable")].astype(np.int32)
    return dicts_from_assignment(strings, cluster, order)


def create_groups(matches: pd.DataFrame, model=None, link_min_similarity: float = 0.75, group_all_strings: bool = False):
    """What `PolyFuzz.group` does per model (reference polyfuzz.py:331-373,459-484): self-match the unique To
    (or From) strings with `model` (default TFIDF(n_gram_range=(3, 3), min_similarity=link_min_similarity)),
    single-linkage them, and add the `Group` column.  Returns (matches with Group, clusters, cluster_mapping).
    With a polyfuzz_amd TFIDF model the self-match result never leaves the device before the linkage."""
    from .models import TFIDF
    if model is None:
        model = TFIDF(n_gram_range=(3, 3), min_similarity=link_min_similarity)
    col = matches.From if group_all_strings else matches.To
    strings = list(col.dropna().unique())
    if isinstance(model, TFIDF) and link_min_similarity >= 0 and len(strings) > 1:
        result = model.match_device(strings)                       # top-1 self-match, diagonal excluded
        clusters, cluster_id_map, cluster_name_map = group_top1(result, strings, link_min_similarity)
    else:
        clusters, cluster_id_map, cluster_name_map = single_linkage(model.match(strings), link_min_similarity)
    df = matches.copy()
    df["Group"] = df["To"].map(cluster_name_map).fillna(df["To"])
    return df, clusters, cluster_id_map
