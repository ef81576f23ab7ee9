"""The two lists PolyFuzz's documentation and benchmarks are defined on, from local files.

Mirrors `polyfuzz.datasets.load_company_names` / `load_movie_titles`
(reference polyfuzz/datasets/_load_data.py:6-40), which download
data/company_names.json and data/movie_titles.json over HTTP.  The GPU box has
no network, so the same two lists ship gzipped in polyfuzz_amd/data/ (inputs of
bench.py and of the parity tests -- data, not code):

* 100 000 SEC-EDGAR company names -- the list the headline metric is defined on
  (`TFIDF(min_similarity=0, top_n=5).match(names)`, docs/tutorial/datasets/datasets.md:36-41);
* {"Netflix": 6 172 titles, "IMDB": 80 852 titles}.

`c2_lists()` / `c3_lists()` are the seeded sub-lists SURVEY.md §8d fixes for configs 2 and 3.
"""
import gzip
import json
import os
from typing import List, Mapping, Tuple

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_cache = {}


def _load(name):
    if name not in _cache:
        with gzip.open(os.path.join(_DIR, name + ".json.gz"), "rt", encoding="utf-8") as f:
            _cache[name] = json.load(f)
    return _cache[name]


def load_company_names() -> List[str]:
    """100 000 company names (a fresh list on every call, as the reference returns)."""
    return list(_load("company_names"))


def load_movie_titles() -> Mapping[str, List[str]]:
    """{"Netflix": [...6172 titles], "IMDB": [...80852 titles]}"""
    return {k: list(v) for k, v in _load("movie_titles").items()}


def c2_lists(n=10_000) -> Tuple[List[str], List[str]]:
    """perm = default_rng(0).permutation(100000); from = names[perm[:n]],
    to = names[perm[n:2n]] (first from-name: 'AITHON OFFSHORE II LTD.')."""
    names = _load("company_names")
    perm = np.random.default_rng(0).permutation(len(names))
    return [names[i] for i in perm[:n]], [names[i] for i in perm[n:2 * n]]


def c3_lists(n=20_000) -> Tuple[List[str], List[str]]:
    """perm = default_rng(0).permutation(80852) over the IMDB titles;
    from = imdb[perm[:n]] (first: 'Polly Blue Eyes'), to = imdb[perm[n:2n]]."""
    imdb = _load("movie_titles")["IMDB"]
    perm = np.random.default_rng(0).permutation(len(imdb))
    return [imdb[i] for i in perm[:n]], [imdb[i] for i in perm[n:2 * n]]
