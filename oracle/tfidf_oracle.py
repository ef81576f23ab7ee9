"""
oracle/tfidf_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy / pure-Python restatement of the reference's TF-IDF vectorisation:

* ``clean_string``   <- reference polyfuzz/models/_tfidf.py:142-146
* ``create_ngrams``  <- reference polyfuzz/models/_tfidf.py:120-139
* ``TfidfOracle``    <- reference polyfuzz/models/_tfidf.py:102-118, i.e.
  scikit-learn's ``TfidfVectorizer(min_df=1, analyzer=callable)`` (third-party,
  scikit_learn>=0.22.2.post1, setup.py:21; 1.7.2 installed here), whose
  algorithm is restated from sklearn/feature_extraction/text.py:
    - vocabulary = distinct analyzer outputs, sorted; column id = rank
      (text.py:1194-1199, 1247-1310)
    - tf = raw count per document (text.py:1247-1310), OOV terms ignored at
      transform time (text.py:1271-1273)
    - idf = ln((1 + n_docs) / (1 + df)) + 1   (smooth_idf, text.py:1662-1679)
    - value = tf * idf, rows L2-normalised sequentially in float64
      (text.py:1716-1722 -> sklearn/utils/sparsefuncs_fast.pyx
      inplace_csr_row_normalize_l2: sum += x*x in index order, sqrt, divide)

Pure Python loops: fine for the <= ~20k strings the tests use.
"""
import re

import numpy as np


def clean_string(string):
    """reference _tfidf.py:142-146: lower(), keep [A-Za-z0-9 ], collapse whitespace, strip."""
    string = re.sub(r'[^A-Za-z0-9 ]+', '', string.lower())
    string = re.sub(r'\s+', ' ', string).strip()
    return string


def create_ngrams(string, n_gram_range=(3, 3), clean=True, remove_space_ngrams=True):
    """reference _tfidf.py:120-139: sliding character windows, both range ends inclusive."""
    if clean:
        string = clean_string(string)
    out = []
    for n in range(n_gram_range[0], n_gram_range[1] + 1):
        for i in range(len(string) - n + 1):
            g = string[i:i + n]
            if remove_space_ngrams and ' ' in g:
                continue
            out.append(g)
    return out


class TfidfOracle:
    def __init__(self, n_gram_range=(3, 3), clean=True, remove_space_ngrams=True):
        self.n_gram_range = tuple(n_gram_range)
        self.clean = clean
        self.remove_space_ngrams = remove_space_ngrams
        self.vocabulary = None      # sorted list of n-grams
        self.index = None           # n-gram -> column id
        self.idf = None             # float64[V]
        self.df = None              # int64[V]
        self.n_docs = 0

    def _analyze(self, s):
        return create_ngrams(s, self.n_gram_range, self.clean, self.remove_space_ngrams)

    def fit(self, docs):
        seen = {}
        for s in docs:
            for g in set(self._analyze(s)):
                seen[g] = seen.get(g, 0) + 1
        if not seen:
            # sklearn text.py:1282-1285
            raise ValueError("empty vocabulary; perhaps the documents only contain stop words")
        self.vocabulary = sorted(seen)
        self.index = {g: i for i, g in enumerate(self.vocabulary)}
        self.df = np.array([seen[g] for g in self.vocabulary], np.int64)
        self.n_docs = len(docs)
        df = self.df.astype(np.float64) + 1.0
        n = float(self.n_docs + 1)
        self.idf = np.log(n / df) + 1.0
        return self

    def transform(self, docs):
        """-> (indptr int64, indices int32, data float64), sorted indices per row."""
        indptr = [0]
        indices = []
        data = []
        for s in docs:
            cnt = {}
            for g in self._analyze(s):
                c = self.index.get(g)
                if c is not None:
                    cnt[c] = cnt.get(c, 0) + 1
            cols = sorted(cnt)
            vals = [float(cnt[c]) * float(self.idf[c]) for c in cols]
            ss = 0.0
            for v in vals:
                ss += v * v
            if ss != 0.0:
                nrm = float(np.sqrt(ss))
                vals = [v / nrm for v in vals]
            indices.extend(cols)
            data.extend(vals)
            indptr.append(len(indices))
        return (np.array(indptr, np.int64), np.array(indices, np.int32), np.array(data, np.float64))
