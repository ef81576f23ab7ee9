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
  EditDistance class run with the restated scorer.  The same holds for the
  RapidFuzz matcher's scorer and for process.extractOne's first-best / cut-off
  rules (restated from rapidfuzz's documentation): PARITY UNPINNED.
* fuzz_scorers.py: the other rapidfuzz.fuzz scorers (partial_ratio, token_set_ratio, token_ratio,
  partial_token_*, WRatio -- the RapidFuzz matcher's default) restated in plain Python from rapidfuzz 3.x's
  published semantics: PARITY UNPINNED, anchored on the values rapidfuzz publishes
  (tests/test_fuzz_oracle_cpu.py).  fuzz_scorers.c is the same statement in plain C (held equal to the Python file
  bit for bit by the same test): the fast checker of K7 at full list sizes and bench.py's CPU arm for RapidFuzz.
* reference_path.py: the reference's own executable TF-IDF path (sklearn
  vectoriser + dense cosine + full sorts + frame), the CPU arm "(i)" of
  bench.py -- pinned cell for cell on frames the reference package produced.
* single_linkage / precision_recall_curve have no restatement here: the
  reference functions are pure Python and importable in the build container, so
  the goldens under tests/golden/group_golden.json are outputs of the reference
  itself (tests/golden/make_golden_group.py).
"""
from .tfidf_oracle import (clean_string, create_ngrams, TfidfOracle)      # noqa: F401
from .tfidf_numpy import TfidfNumpyOracle      # noqa: F401
from .dense import dense_cossim, dense_cossim_topn   # noqa: F401
from .native import (cossim_topn, cossim_dense, indel_ratio, indel_argmax, fuzz_score, fuzz_extract_one, fuzz_matrix,  # noqa: F401
                     build as build_native)
