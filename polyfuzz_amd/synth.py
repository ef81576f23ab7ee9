"""Seeded synthetic company-name-like string lists (benchmark / test inputs).

The reference's datasets are HTTP downloads (polyfuzz/datasets/_load_data.py:6-40)
and do not exist on the GPU box; this generator recombines the token statistics
of the real company-name list (polyfuzz_amd/data/company_tokens.json.gz, made by
tools/make_token_table.py) so that the character-3-gram distribution -- Zipfian,
with the heavy 'inc' / 'llc' posting lists that dominate the sparse product --
matches the real data.
"""
import gzip
import json
import os

import numpy as np

_TABLE = None


def _table():
    global _TABLE
    if _TABLE is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "company_tokens.json.gz")
        with gzip.open(path, "rt", encoding="utf-8") as f:
            t = json.load(f)
        tokens = np.array(t["tokens"], dtype=object)
        p = np.array(t["token_counts"], np.float64)
        ks = np.array([k for k, _ in t["tokens_per_name"]], np.int64)
        kp = np.array([c for _, c in t["tokens_per_name"]], np.float64)
        _TABLE = (tokens, p / p.sum(), ks, kp / kp.sum())
    return _TABLE


def company_names(n, seed):
    """n synthetic names, deterministic in (n, seed)."""
    tokens, p, ks, kp = _table()
    rng = np.random.default_rng(seed)
    k = rng.choice(ks, size=n, p=kp)
    k = np.maximum(k, 1)
    flat = rng.choice(len(tokens), size=int(k.sum()), p=p)
    words = tokens[flat]
    out = []
    pos = 0
    for ki in k.tolist():
        out.append(" ".join(words[pos:pos + ki]))
        pos += ki
    return out
