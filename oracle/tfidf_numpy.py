"""
oracle/tfidf_numpy.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The SAME restatement as oracle/tfidf_oracle.py (reference polyfuzz/models/_tfidf.py:102-146 -> scikit-learn's
TfidfVectorizer(min_df=1, analyzer=...), sklearn/feature_extraction/text.py:1194-1199, 1247-1310, 1662-1679, 1716-1722),
vectorised with numpy so that it finishes in seconds on a MILLION strings -- the Python-loop restatement takes ~20 us per
string and pass, and bench.py's 1M-shard configuration (1 125 000 strings) used to skip the vectoriser check for that reason
(VERDICT r4 weak 1c).  Covers what that configuration and the headline run: clean_string=True, remove_space_ngrams=True, any
n_gram_range whose upper end fits a 63-bit code (n <= 11).

PINNED on the Python restatement bit for bit (tests/test_oracle_cpu.py::test_numpy_vectoriser_equals_the_python_one_bitwise),
which is pinned on scikit-learn bit for bit: same vocabulary order, same float64 idf, same sequential sum of squares.

How each sklearn rule is kept:
  * vocabulary = distinct n-grams in Python string order: an n-gram is coded big-endian in base 38 (0 = pad, 1 = ' ',
    2..11 = '0'..'9', 12..37 = 'a'..'z': ASCII order), left-aligned to the longest n -- numeric order == string order;
  * tf = occurrences per document, df = documents per n-gram: one sort of (document, n-gram) pairs;
  * idf = ln((1 + n_docs) / (1 + df)) + 1 in float64: the same numpy expression as the Python restatement;
  * row normalisation: sum += x * x IN INDEX ORDER, sqrt, divide -- rows are grouped by their number of entries and the
    sum runs column by column over each group (numpy's own reductions are pairwise and round differently).
"""
import re

import numpy as np

_BASE = 38


def _clean_all(strings):
    """reference _tfidf.py:142-146 for every string: lower(), keep [A-Za-z0-9 ], collapse whitespace, strip"""
    drop = re.compile(r'[^A-Za-z0-9 ]+')
    ws = re.compile(r'\s+')
    return [ws.sub(' ', drop.sub('', s.lower())).strip() for s in strings]


def _pack(cleaned):
    lens = np.fromiter((len(s) for s in cleaned), np.int64, len(cleaned))
    off = np.zeros(len(cleaned) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    raw = np.frombuffer("".join(cleaned).encode("ascii"), np.uint8)
    return raw, off


def _ranks(raw):
    r = np.zeros(len(raw), np.int64)
    sp = raw == 32
    dg = (raw >= 48) & (raw <= 57)
    lt = (raw >= 97) & (raw <= 122)
    r[sp] = 1
    r[dg] = raw[dg].astype(np.int64) - 48 + 2
    r[lt] = raw[lt].astype(np.int64) - 97 + 12
    assert (sp | dg | lt).all(), "cleaned text holds only [a-z0-9 ]"
    return r, sp


def _ngram_codes(raw, off, lo, hi):
    """(document of every kept n-gram, its code), documents ascending, in the analyzer's order inside a document
    (all n = lo first, then lo + 1 ...: reference _tfidf.py:130-139 -- the order only matters for nothing here: counts)"""
    assert hi <= 11, "codes of up to 11 characters fit 63 bits"
    n_docs = len(off) - 1
    ranks, is_sp = _ranks(raw)
    doc_of = np.repeat(np.arange(n_docs, dtype=np.int64), np.diff(off))
    end_of = off[1:][doc_of] if len(raw) else np.zeros(0, np.int64)
    spaces = np.concatenate([[0], np.cumsum(is_sp, dtype=np.int64)])
    pos = np.arange(len(raw), dtype=np.int64)
    docs, codes = [], []
    for n in range(lo, hi + 1):
        ok = pos + n <= end_of
        p = pos[ok]
        p = p[spaces[p + n] - spaces[p] == 0]             # remove_space_ngrams: windows holding a ' ' are dropped
        c = np.zeros(len(p), np.int64)
        for k in range(n):
            c = c * _BASE + ranks[p + k]
        c *= _BASE ** (hi - n)                            # left-aligned: 'ab' < 'aba' like the strings
        docs.append(doc_of[p])
        codes.append(c)
    return np.concatenate(docs), np.concatenate(codes)


def _decode(code, hi):
    out = []
    for k in range(hi):
        d = (code // _BASE ** (hi - 1 - k)) % _BASE
        if d == 0:
            break
        out.append(' ' if d == 1 else (chr(48 + d - 2) if d < 12 else chr(97 + d - 12)))
    return "".join(out)


class TfidfNumpyOracle:
    """fit(docs) / transform(docs) like oracle.TfidfOracle; transform returns (indptr int64, indices int32, data float64)"""

    def __init__(self, n_gram_range=(3, 3)):
        self.lo, self.hi = int(n_gram_range[0]), int(n_gram_range[1])
        self.codes = None        # sorted distinct codes == the vocabulary in column order
        self.df = self.idf = None
        self.n_docs = 0

    @property
    def vocabulary(self):
        return [_decode(int(c), self.hi) for c in self.codes]

    def _counts(self, docs):
        raw, off = _pack(_clean_all(docs))
        d, c = _ngram_codes(raw, off, self.lo, self.hi)
        return len(off) - 1, d, c

    def fit(self, docs):
        n_docs, d, c = self._counts(docs)
        if len(c) == 0:
            raise ValueError("empty vocabulary; perhaps the documents only contain stop words")     # sklearn text.py:1282-1285
        self.codes = np.unique(c)
        col = np.searchsorted(self.codes, c)
        pairs = np.unique(d * len(self.codes) + col)                   # one entry per (document, n-gram)
        self.df = np.bincount(pairs % len(self.codes), minlength=len(self.codes)).astype(np.int64)
        self.n_docs = n_docs
        self.idf = np.log(float(n_docs + 1) / (self.df.astype(np.float64) + 1.0)) + 1.0
        self._fitted = (d, col)                                        # for transform_fitted(): no second cleaning pass
        return self

    def transform_fitted(self, lo, hi):
        """transform(docs[lo:hi]) of the list fit() saw, from the n-grams fit() already extracted"""
        d, col = self._fitted
        a, b = np.searchsorted(d, [lo, hi])                            # documents ascend inside each n; one n ...
        if self.lo != self.hi:                                         # ... several: select by mask instead
            m = (d >= lo) & (d < hi)
            return self._rows(hi - lo, d[m] - lo, col[m])
        return self._rows(hi - lo, d[a:b] - lo, col[a:b])

    def transform(self, docs):
        n_docs, d, c = self._counts(docs)
        v = len(self.codes)
        col = np.searchsorted(self.codes, c)
        known = (col < v)
        known[known] &= self.codes[col[known]] == c[known]            # out-of-vocabulary n-grams are ignored (text.py:1271-1273)
        return self._rows(n_docs, d[known], col[known])

    def _rows(self, n_docs, d, col):
        v = len(self.codes)
        key, tf = np.unique(d * v + col, return_counts=True)           # sorted by (document, column): CSR order
        rows, cols = key // v, (key % v).astype(np.int32)
        indptr = np.zeros(n_docs + 1, np.int64)
        np.cumsum(np.bincount(rows, minlength=n_docs), out=indptr[1:])
        data = tf.astype(np.float64) * self.idf[cols]
        # L2 rows, the sum of squares in index order (sparsefuncs_fast.pyx inplace_csr_row_normalize_l2)
        nnz_row = np.diff(indptr)
        norm = np.ones(n_docs, np.float64)
        for length in np.unique(nnz_row):
            if length == 0:
                continue
            r = np.nonzero(nnz_row == length)[0]
            base = indptr[r]
            ss = np.zeros(len(r), np.float64)
            for k in range(int(length)):
                x = data[base + k]
                ss += x * x
            norm[r] = np.sqrt(ss)
        norm[norm == 0.0] = 1.0
        data = data / np.repeat(norm, nnz_row)
        return indptr, cols, data
