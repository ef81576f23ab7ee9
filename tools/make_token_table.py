"""Derive the token-frequency table used by the synthetic company-name generator
(polyfuzz_amd/synth.py) from the reference's data/company_names.json.  Run in the
build container only (the GPU box has no /root/reference); the output is committed.

Recipe: a synthetic name is k whitespace tokens drawn
by frequency from the real names' tokens, k drawn from the real tokens-per-name
distribution -- this keeps the Zipfian 3-gram distribution (heavy 'inc', 'llc').
"""
import collections
import gzip
import json
import os

REF = "/root/reference/data/company_names.json"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "polyfuzz_amd", "data", "company_tokens.json.gz")

names = json.load(open(REF))
tok = collections.Counter(t for s in names for t in s.split())
cnt = collections.Counter(len(s.split()) for s in names)
items = sorted(tok.items(), key=lambda kv: (-kv[1], kv[0]))
table = {
    "source": "token statistics of MaartenGr/PolyFuzz data/company_names.json (100000 names)",
    "tokens": [t for t, _ in items],
    "token_counts": [c for _, c in items],
    "tokens_per_name": sorted(cnt.items()),
}
with gzip.open(OUT, "wt", encoding="utf-8") as f:
    json.dump(table, f)
print(OUT, os.path.getsize(OUT), "bytes;", len(items), "distinct tokens")
