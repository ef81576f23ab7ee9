"""
oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's hot path (PolyFuzz v0.4.3: TF-IDF
char-n-gram cosine top-n, and the all-pairs Indel-ratio arg-max).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / the timed CPU baseline
-- never as a compute path of ``polyfuzz_amd``.

Pinning status (see DESIGN.md "Oracle"):
* TF-IDF vectorisation: pinned bit-for-bit against scikit-learn's
  TfidfVectorizer (installed here and on the GPU box) and against the
  reference run in this container (tests/golden/, made by
  tests/golden/make_golden.py), incl. the README known answers.
* cosine top-n: pinned against the reference's own "sklearn" back-end run in
  this container (tests/golden/); sparse_dot_topn itself is not installed, its
  strict ``> lower_bound`` / top-n semantics are restated from its published
  behaviour.
* Indel ratio (rapidfuzz.fuzz.ratio): rapidfuzz is not installed anywhere we
  can run -> PARITY UNPINNED for the scorer; plumbing (arg-max, self-match
  removal, normalisation) is pinned through the reference's own
  EditDistance class run with the restated scorer.
"""
from .tfidf_oracle import (clean_string, create_ngrams, TfidfOracle)      # noqa: F401
from .dense import dense_cossim, dense_cossim_topn   # noqa: F401
from .native import (cossim_topn, cossim_dense, indel_ratio, indel_argmax,  # noqa: F401
                     build as build_native)
