"""GPU parity of K3 (sparse cosine top-n) against the oracle and the reference goldens.

Bar: indices bit-exact against the oracle's canonical order (score desc, col
asc) except where the float64 oracle itself separates two candidates by less
than NEAR_TIE (fp32 accumulation cannot resolve those); scores within 1e-5.
"""
import numpy as np
import pytest

from tests.helpers import assert_topn_parity, vectorize_pair, random_csr

pytestmark = pytest.mark.gpu


def _run(ctx, a3, b3, n_col, ntop, lb, diag=False):
    from polyfuzz_amd import _lib
    return _lib.cossim_topn_host(ctx, a3, b3, n_col, ntop, lb, diag)


def test_readme_lists(ctx, oracle_mod):
    fl = ["apple", "apples", "appl", "recal", "house", "similarity"]
    tl = ["apple", "apples", "mouse"]
    a3, b3, n_col = vectorize_pair(oracle_mod, fl, tl)
    idx, val = _run(ctx, a3, b3, n_col, 3, 0.0)
    # known answers: SURVEY.md §8c / reference README.md:88-95
    np.testing.assert_array_equal(idx[:, 0], [0, 1, 0, -1, 2, -1])
    np.testing.assert_allclose(val[:, 0], [1.0, 1.0, 0.783751479, 0.0, 0.587926596, 0.0], atol=1e-6)
    np.testing.assert_array_equal(idx[:, 1], [1, 0, 1, -1, -1, -1])
    np.testing.assert_allclose(val[:3, 1], [0.777651595, 0.777651595, 0.609485588], atol=1e-6)
    assert (idx[:, 2] == -1).all()
    # strict lower bound 0.75: house->mouse (0.588) disappears
    idx, val = _run(ctx, a3, b3, n_col, 1, 0.75)
    np.testing.assert_array_equal(idx[:, 0], [0, 1, 0, -1, -1, -1])


@pytest.mark.parametrize("ntop,lb", [(5, 0.0), (1, 0.0), (5, 0.75), (10, 0.3)])
def test_company_c2_vs_oracle(ctx, oracle_mod, golden, ntop, lb):
    fl = golden["company_c2_lists"]["from_list"]
    tl = golden["company_c2_lists"]["to_list"]
    a3, b3, n_col = vectorize_pair(oracle_mod, fl, tl, cache_key="c2")
    idx, val = _run(ctx, a3, b3, n_col, ntop, lb)
    exp_idx, exp_val = oracle_mod.cossim_topn(a3, b3, n_col, ntop, lb)
    assert_topn_parity(idx, val, exp_idx, exp_val, oracle_mod, a3, b3, n_col)


def test_company_c2_vs_reference_golden(ctx, oracle_mod, golden):
    """Against the reference's own run (tests/golden/make_golden.py): canonical top-5 of its dense matrix."""
    fl = golden["company_c2_lists"]["from_list"]
    tl = golden["company_c2_lists"]["to_list"]
    a3, b3, n_col = vectorize_pair(oracle_mod, fl, tl, cache_key="c2")
    assert n_col == int(golden["npz"]["c2_vocab_size"][0])
    idx, val = _run(ctx, a3, b3, n_col, 5, 0.0)
    assert_topn_parity(idx, val, golden["npz"]["c2_canon_idx"], golden["npz"]["c2_canon_val"],
                       oracle_mod, a3, b3, n_col)
    # the reference's printed (3-dp rounded) scores
    np.testing.assert_allclose(np.round(val.astype(np.float64), 3), golden["npz"]["c2_ref_sim"], atol=1.01e-3)


def test_company_self_match(ctx, oracle_mod, golden):
    sl = golden["company_self_list"]["from_list"]
    a3, b3, n_col = vectorize_pair(oracle_mod, sl, None)
    idx, val = _run(ctx, a3, a3, n_col, 3, 0.0, diag=True)
    exp_idx, exp_val = oracle_mod.cossim_topn(a3, a3, n_col, 3, 0.0, exclude_diag=True)
    assert_topn_parity(idx, val, exp_idx, exp_val, oracle_mod, a3, a3, n_col, exclude_diag=True)
    assert (idx[:, 0] != np.arange(len(sl))).all()
    assert_topn_parity(idx, val, golden["npz"]["self_canon_idx"], golden["npz"]["self_canon_val"],
                       oracle_mod, a3, a3, n_col, exclude_diag=True)


@pytest.mark.parametrize("n_a,n_b,n_col,dens,ntop", [
    (1, 1, 5, 0.9, 1), (3, 5000, 40, 0.2, 7), (257, 2049, 300, 0.05, 4), (64, 4097, 64, 0.5, 128),
    (100, 70, 2000, 0.09, 3),   # rows with > 64 n-grams
    (40, 3000, 50, 0.4, 300), (9, 5000, 30, 0.6, 1024),   # top_n beyond 128: the 1152-key candidate buffer
])
def test_random_csr_shapes(ctx, oracle_mod, n_a, n_b, n_col, dens, ntop):
    rng = np.random.default_rng(n_a * 7 + n_b)
    a3 = random_csr(rng, n_a, n_col, dens)
    b3 = random_csr(rng, n_b, n_col, dens)
    idx, val = _run(ctx, a3, b3, n_col, ntop, 0.1)
    exp_idx, exp_val = oracle_mod.cossim_topn(a3, b3, n_col, ntop, 0.1)
    assert_topn_parity(idx, val, exp_idx, exp_val, oracle_mod, a3, b3, n_col)


def test_empty_and_degenerate(ctx, oracle_mod):
    rng = np.random.default_rng(3)
    b3 = random_csr(rng, 50, 30, 0.2)
    # from-side with empty rows
    a3 = random_csr(rng, 20, 30, 0.2, empty_rows=[0, 7, 19])
    idx, val = _run(ctx, a3, b3, 30, 2, 0.0)
    assert (idx[[0, 7, 19]] == -1).all() and (val[[0, 7, 19]] == 0).all()
    exp_idx, exp_val = oracle_mod.cossim_topn(a3, b3, 30, 2, 0.0)
    assert_topn_parity(idx, val, exp_idx, exp_val, oracle_mod, a3, b3, 30)
    # empty to-side
    e3 = (np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float64))
    idx, val = _run(ctx, a3, e3, 30, 2, 0.0)
    assert idx.shape == (20, 2) and (idx == -1).all()
    # empty from-side
    idx, val = _run(ctx, e3, b3, 30, 2, 0.0)
    assert idx.shape == (0, 2)
    # duplicated to-rows: exact ties resolve to the lower column
    dup = (np.concatenate([b3[0], b3[0][1:] + b3[0][-1]]), np.concatenate([b3[1], b3[1]]),
           np.concatenate([b3[2], b3[2]]))
    idx, val = _run(ctx, a3, dup, 30, 4, 0.0)
    exp_idx, exp_val = oracle_mod.cossim_topn(a3, dup, 30, 4, 0.0)
    np.testing.assert_array_equal(idx, exp_idx)


def test_index_build_paths_agree(ctx, oracle_mod, monkeypatch):
    """The LDS-histogram index build and the global-atomic one kept for huge vocabularies
    (PFZ_NO_LDS_HIST) lead to bit-identical K3 results (integer sums do not depend on the order of the
    postings inside a list), with to-rows spread over several to-blocks."""
    rng = np.random.default_rng(11)
    a3 = random_csr(rng, 300, 501, 0.03)
    b3 = random_csr(rng, 7001, 501, 0.03)
    idx, val = _run(ctx, a3, b3, 501, 8, 0.05)
    exp_idx, exp_val = oracle_mod.cossim_topn(a3, b3, 501, 8, 0.05)
    assert_topn_parity(idx, val, exp_idx, exp_val, oracle_mod, a3, b3, 501)
    monkeypatch.setenv("PFZ_NO_LDS_HIST", "1")
    idx2, val2 = _run(ctx, a3, b3, 501, 8, 0.05)
    np.testing.assert_array_equal(idx2, idx)
    np.testing.assert_array_equal(val2, val)


@pytest.mark.parametrize("knobs", [
    {"PFZ_K3_BLOCK": "4096"}, {"PFZ_K3_BLOCK": "1024"}, {"PFZ_K3_BLOCK": "1536"},
    {"PFZ_K3_SLICES": "3"}, {"PFZ_K3_SLICES": "7", "PFZ_K3_BLOCK": "1024"},
    {"PFZ_K3_BANK_ORDER": "1"}, {"PFZ_K3_BANK_ORDER": "1", "PFZ_K3_BLOCK": "4096"}, {"PFZ_K3_NO_BANK_ORDER": "1"},
])
def test_tuning_knobs_do_not_change_results(ctx, oracle_mod, monkeypatch, knobs):
    """Every launch shape the tuning knobs can select (to-block size, to-side slices) gives the default
    shape's results bit for bit -- and the oracle's."""
    rng = np.random.default_rng(17)
    a3 = random_csr(rng, 260, 400, 0.04)
    b3 = random_csr(rng, 20000, 400, 0.04)
    ref_idx, ref_val = _run(ctx, a3, b3, 400, 6, 0.1)
    exp_idx, exp_val = oracle_mod.cossim_topn(a3, b3, 400, 6, 0.1)
    assert_topn_parity(ref_idx, ref_val, exp_idx, exp_val, oracle_mod, a3, b3, 400)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    idx, val = _run(ctx, a3, b3, 400, 6, 0.1)
    np.testing.assert_array_equal(idx, ref_idx)
    np.testing.assert_array_equal(val, ref_val)


def test_bad_arguments(ctx):
    from polyfuzz_amd import _lib, PfzError
    e3 = (np.zeros(2, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32))
    with pytest.raises(PfzError):
        _lib.cossim_topn_host(ctx, e3, e3, 4, 0, 0.0)
    idx, val = _lib.cossim_topn_host(ctx, e3, e3, 4, 1025, 0.0)      # beyond 1024: passes of 1024 (no limit any more)
    assert idx.shape == (1, 1025) and (idx == -1).all() and (val == 0).all()


@pytest.mark.parametrize("ntop,lb,diag", [(1500, 0.0, False), (2300, 0.0, True), (1025, 0.1, False), (2048, 0.0, False)])
def test_deep_top_n(ctx, oracle_mod, ntop, lb, diag):
    """top_n beyond the 1024 keys one pass keeps (the reference clips top_n to the number of distinct to-strings only,
    _utils.py:54-56): passes of 1024, each continuing strictly below the last key of the pass before -- against the oracle,
    with exact ties across the pass boundary (duplicate to-rows), rows that run out of positive sums before top_n, a lower
    bound, and the diagonal left out."""
    rng = np.random.default_rng(314)
    n_col = 60                                       # few columns: most pairs share one, rows have > 2000 positive sums
    a3 = random_csr(rng, 300, n_col, 0.08, empty_rows=(4,))
    b3 = random_csr(rng, 2600, n_col, 0.08, empty_rows=(7, 1000))
    if diag:
        a3 = b3
    else:
        # duplicate to-rows: equal sums in long runs, the column index decides -- also where a pass ends
        bp, bi, bv = [np.array(x) for x in b3]
        rows = [(bi[bp[r]:bp[r + 1]], bv[bp[r]:bp[r + 1]]) for r in range(len(bp) - 1)]
        for r in range(50, 2600, 2):
            rows[r] = rows[r % 40]
        ptr = np.zeros(len(rows) + 1, np.int64)
        for r, (c, _) in enumerate(rows):
            ptr[r + 1] = ptr[r] + len(c)
        b3 = (ptr, np.concatenate([c for c, _ in rows]).astype(np.int32), np.concatenate([v for _, v in rows]).astype(np.float64))
    idx, val = _run(ctx, a3, b3, n_col, ntop, lb, diag)
    exp_idx, exp_val = oracle_mod.cossim_topn(a3, b3, n_col, ntop, lb, exclude_diag=diag)
    assert_topn_parity(idx, val, exp_idx, exp_val, oracle_mod, a3, b3, n_col, exclude_diag=diag)
    # scores descend across the pass boundaries too (the order of exact ties -- by column -- is part of the parity check above;
    # two DIFFERENT integer sums may round to the same fp32 score, so equal scores alone say nothing about the columns)
    assert (np.diff(val.astype(np.float64), axis=1) <= 0).all()
    for r in range(len(idx)):                        # no column twice in a row
        got = idx[r][idx[r] >= 0]
        assert len(set(got.tolist())) == len(got)
    assert ((idx >= 0).sum(axis=1) < ntop).any()     # some rows do run out


def test_tfidf_match_top_n_beyond_1024(ctx):
    """TFIDF(top_n=1100).match on 1 300 real names: the frame has 1 + 2 * 1100 columns (top_n clipped to the distinct to-strings
    as in the reference), its first columns equal the top-5 frame's."""
    from polyfuzz_amd import datasets
    from polyfuzz_amd.models import TFIDF
    names = datasets.load_company_names()[:1300]
    deep = TFIDF(min_similarity=0.0, top_n=1100).match(names[:200], names)
    top5 = TFIDF(min_similarity=0.0, top_n=5).match(names[:200], names)
    assert deep.shape == (200, 1 + 2 * 1100)
    for c in top5.columns:
        assert deep[c].tolist() == top5[c].tolist(), c


@pytest.mark.parametrize("fa,fb", [(7.5, 3.0), (1e-3, 2e-2), (300.0, 1e-4)])
def test_unnormalised_rows_scale_safely(ctx, oracle_mod, fa, fb):
    """The operator is also called on matrices that are NOT L2-normalised (reference _utils.py:74-82: the
    sparse branch returns raw dot products).  K3 sizes its fixed-point scale from the largest row norms,
    so scores keep ~2^-30 relative precision and cannot overflow."""
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(5)
    a3 = random_csr(rng, 90, 120, 0.15)
    b3 = random_csr(rng, 700, 120, 0.15)
    sa = fa * (0.5 + rng.random(len(a3[2])))
    sb = fb * (0.5 + rng.random(len(b3[2])))
    a3 = (a3[0], a3[1], a3[2] * sa)
    b3 = (b3[0], b3[1], b3[2] * sb)
    idx, val = _lib.cossim_topn_host(ctx, a3, b3, 120, 6, 0.0)
    e_idx, e_val = oracle_mod.cossim_topn((a3[0], a3[1], a3[2].astype(np.float32).astype(np.float64)),
                                          (b3[0], b3[1], b3[2].astype(np.float32).astype(np.float64)), 120, 6, 0.0)
    bound = fa * fb * 2.5
    np.testing.assert_allclose(val, e_val, rtol=0, atol=1e-5 * bound)
    assert (idx != e_idx).any(axis=1).mean() < 0.05
    assert (val >= 0).all()


def test_padded_index_layout(ctx):
    """The inverted index stores every (n-gram, to-block) list padded to whole 16-posting pieces: the piece
    count is what the list lengths say, whichever build path made it."""
    import os
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(23)
    b3 = random_csr(rng, 5000, 300, 0.05)
    for env in (None, "1"):
        if env:
            os.environ["PFZ_NO_LDS_HIST"] = env
        try:
            m = _lib.DeviceCSR.upload(ctx, b3[0], b3[1], b3[2].astype(np.float32), 300)
            info = _lib.DeviceIndex.build(ctx, m).info()
        finally:
            os.environ.pop("PFZ_NO_LDS_HIST", None)
        blk = info["block_cols"]
        rows = np.repeat(np.arange(5000), np.diff(b3[0]))
        cnt = np.bincount(b3[1].astype(np.int64) * info["n_blocks"] + rows // blk, minlength=300 * info["n_blocks"])
        assert info["piece_postings"] == 16 and info["n_pieces"] == int(((cnt + 15) // 16).sum())


@pytest.mark.parametrize("shape", ["many_lists", "long_lists", "more_than_64_ngrams"])
def test_scatter_round_shapes(ctx, oracle_mod, shape):
    """The scatter deals the pieces of a from-row's lists 64 per round, 4 per step: rows whose blocks need
    several rounds (> 64 pieces), lists of hundreds of postings (one n-gram shared by most to-rows), and
    from-rows with more than 64 n-grams (second pass over the lanes) against the oracle."""
    rng = np.random.default_rng({"many_lists": 1, "long_lists": 2, "more_than_64_ngrams": 3}[shape])
    n_col = 200
    if shape == "many_lists":
        a3 = random_csr(rng, 50, n_col, 0.3)            # ~60 n-grams per from-row
        b3 = random_csr(rng, 9000, n_col, 0.1)          # every list ~200 postings per block: ~13 pieces x 60 lists
    elif shape == "long_lists":
        a3 = random_csr(rng, 40, n_col, 0.04)
        b3 = random_csr(rng, 6000, n_col, 0.6)          # lists of ~1200 postings per block
    else:
        a3 = random_csr(rng, 30, n_col, 0.75)           # ~150 n-grams per from-row
        b3 = random_csr(rng, 3000, n_col, 0.08)
    idx, val = _run(ctx, a3, b3, n_col, 7, 0.0)
    exp_idx, exp_val = oracle_mod.cossim_topn(a3, b3, n_col, 7, 0.0)
    assert_topn_parity(idx, val, exp_idx, exp_val, oracle_mod, a3, b3, n_col)


@pytest.mark.parametrize("knobs", [
    {"PFZ_K3_LS_BLOCKS": "1"}, {"PFZ_K3_LS_BLOCKS": "2"}, {"PFZ_K3_LS_BLOCKS": "4"},
    {"PFZ_K3_BLOCK": "4096", "PFZ_K3_LS_BLOCKS": "1"}, {"PFZ_K3_BLOCK": "4096", "PFZ_K3_LS_BLOCKS": "2"},
    {"PFZ_K3_LS_BLOCKS": "1", "PFZ_K3_LS_WAVES": "2"}, {"PFZ_K3_LS_BLOCKS": "8", "PFZ_K3_LS_CHUNK": "5"},
    {"PFZ_K3_LS_BLOCKS": "2", "PFZ_K3_LS_CHUNK": "1"},
])
@pytest.mark.parametrize("ntop,lb,diag", [(5, 0.0, False), (1, 0.0, True), (10, 0.2, False), (32, 0.0, True)])
def test_lockstep_kernel_equals_the_row_major_kernel(ctx, oracle_mod, monkeypatch, knobs, ntop, lb, diag):
    """k3_lockstep.hip (to-block-major: slices of to-blocks outer, batches of four from-rows inner, the rows' top-n state
    carried through HBM between slices) against the main kernel on the same index -- bit for bit, every slice width and
    block size, with and without the diagonal -- and against the oracle.  The lists: 3 000 from-rows incl. empty rows, rows
    of more than 64 n-grams (the slow loop) and batches whose four rows do not fit 64 lanes (several windows); 9 000
    to-rows = 5 / 3 blocks."""
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(41)
    n_col = 700
    a3 = random_csr(rng, 3000, n_col, 0.02, empty_rows=(0, 5, 6, 7, 8, 2999))
    # a few heavy from-rows: > 64 entries, and runs of 20..30-entry rows (four of them overflow one window)
    ip, ix_, dv = [np.array(x) for x in a3]
    heavy = random_csr(rng, 40, n_col, 0.12)
    mid = random_csr(rng, 200, n_col, 0.035)
    def cat(parts):
        ptr = [0]
        for p in parts:
            ptr.extend((np.asarray(p[0][1:]) + ptr[-1]).tolist())
        return (np.array(ptr, np.int64), np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts]))
    a3 = cat([(ip, ix_, dv), heavy, mid])
    n_b = len(a3[0]) - 1 if diag else 9000
    b3 = a3 if diag else random_csr(rng, n_b, n_col, 0.02)
    for k, v in knobs.items():
        if k == "PFZ_K3_BLOCK":
            monkeypatch.setenv(k, v)
    monkeypatch.setenv("PFZ_K3_LOCKSTEP", "0")
    ref_idx, ref_val = _run(ctx, a3, b3, n_col, ntop, lb, diag)
    exp_idx, exp_val = oracle_mod.cossim_topn(a3, b3, n_col, ntop, lb, exclude_diag=diag)
    assert_topn_parity(ref_idx, ref_val, exp_idx, exp_val, oracle_mod, a3, b3, n_col, exclude_diag=diag)
    monkeypatch.setenv("PFZ_K3_LOCKSTEP", "1")
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for _ in range(2):              # twice: the pull counters and the state scratch are re-used
        idx, val = _run(ctx, a3, b3, n_col, ntop, lb, diag)
        np.testing.assert_array_equal(idx, ref_idx)
        np.testing.assert_array_equal(val, ref_val)


def test_sparse_dot_topn_pin_fixture(ctx):
    """K3 against tests/golden/sparse_dot_topn_pin.json (pin_sparse_dot_topn.py: `awesome_cossim_topn(A, B.T, ntop,
    lower_bound)` on the README lists and 300 x 291 real company names -- the library's own rows once the script has run where
    it is importable, the oracle's until then): kept columns per row equal except where the fixture's own scores are within
    2e-6 of each other or of the bound (fp32 cannot order those), scores within 1e-5."""
    import importlib
    import json
    import os
    import sys
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    pin = json.load(open(os.path.join(g, "sparse_dot_topn_pin.json"), encoding="utf-8"))
    sys.path.insert(0, g)
    mod = importlib.import_module("pin_sparse_dot_topn")
    t = lambda m: (m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float64))
    for (name, fl, tl, a, b, ntop, lb), rec in zip(mod.cases(), pin["cases"]):
        idx, val = _run(ctx, t(a), t(b), a.shape[1], ntop, lb)
        soft = 0
        for r, exp in enumerate(rec["rows"]):
            got = [(int(j), float(v)) for j, v in zip(idx[r], val[r]) if j >= 0]
            e_cols, e_vals = [c for c, _ in exp], [v for _, v in exp]
            if [c for c, _ in got] != e_cols:
                near = any(abs(x - y) < 2e-6 for x, y in zip(e_vals, e_vals[1:])) or any(abs(v - lb) < 2e-6 for v in e_vals) or \
                    any(abs(v - lb) < 2e-6 for _, v in got)
                assert near, (name, ntop, lb, r, got, exp)
                soft += 1
                continue
            assert all(abs(v - w) <= 1e-5 for (_, v), w in zip(got, e_vals))
        assert soft <= 3, (name, ntop, lb, soft)


def _sym_lists(rng, n, n_col, weird=True):
    """A self-match list of n rows over n_col (+2) columns for the symmetric kernel's tests: random rows incl. empty ones,
    rows of more than 64 n-grams, runs of DUPLICATE rows (exact ties across blocks), and -- `weird` -- two rows of block 3
    that share their only n-gram with 600 rows of block 0 each and with nobody of their own block (see below)."""
    base = random_csr(rng, n, n_col, 0.02, empty_rows=(0, 7, 2048, n - 1))
    ip, ix_, dv = [np.array(x) for x in base]
    rows = [(ix_[ip[r]:ip[r + 1]].copy(), dv[ip[r]:ip[r + 1]].copy()) for r in range(n)]
    heavy = random_csr(rng, 30, n_col, 0.12)
    for k in range(30):                                  # rows of > 64 n-grams, spread over the blocks
        r = (k * 293 + 11) % n
        rows[r] = (heavy[1][heavy[0][k]:heavy[0][k + 1]].copy(), heavy[2][heavy[0][k]:heavy[0][k + 1]].copy())
    for src in (100, 2500, 5000):                        # duplicates of one row in every block: exact ties, column order decides
        for r in range(src + 1, n, 997):
            rows[r] = (rows[src][0].copy(), rows[src][1].copy())
    if weird:
        # two rows of block 3, of ONE LDS bank class (17 and 17 + 32), each with a single n-gram of its own (extra columns) that 600
        # rows of block 0 share and nobody of block 3: both keep threshold 0 in the own-block pass.  The lower one is its class'
        # "magnet" (it fetches its matches below its own block itself), the other is sent 600 candidates for 512 push slots and
        # is recomputed in full
        for w, (sp, first) in enumerate(((n_col, 5), (n_col + 1, 6))):
            for r in range(first, first + 2 * 600, 2):
                c, v = rows[r]
                c, v = np.append(c, sp).astype(np.int32), np.append(v * 0.8, 0.6)
                rows[r] = (c, v / np.sqrt((v * v).sum()))
            rows[3 * 2048 + 17 + 32 * w] = (np.array([sp], np.int32), np.array([1.0]))
    ptr = np.zeros(n + 1, np.int64)
    for r in range(n):
        ptr[r + 1] = ptr[r] + len(rows[r][0])
    return (ptr, np.concatenate([c for c, _ in rows]).astype(np.int32), np.concatenate([v for _, v in rows]).astype(np.float64))


def _self_match(ctx, a3, n_col, ntop, lb, ranges=None, repeats=1, census=None):
    """device-level self-match (one matrix, its own index: what TFIDF.match(list) enqueues), whole or in row ranges"""
    from polyfuzz_amd import _lib
    csr = _lib.DeviceCSR.upload(ctx, a3[0], a3[1], a3[2], n_col)
    index = _lib.DeviceIndex.build(ctx, csr)
    n = len(a3[0]) - 1
    out = None
    for _ in range(repeats):
        if ranges is None:
            out = _lib.cossim_topn(ctx, index, csr, ntop, lb, exclude_diag=True, out=out)
        else:
            for lo, hi in ranges:
                out = _lib.cossim_topn(ctx, index, csr, ntop, lb, exclude_diag=True, out=out, rows=(lo, hi))
    if census is not None:
        census.append(index.symmetric_census())
    return out.download()


@pytest.mark.parametrize("ntop,lb", [(5, 0.0), (1, 0.0), (10, 0.2), (32, 0.0)])
def test_symmetric_kernel_equals_the_row_major_kernel(ctx, oracle_mod, monkeypatch, ntop, lb):
    """k3_symmetric.hip (every unordered pair of rows scored once, candidates handed to the higher row through HBM) against
    the row-major kernel on the same matrix -- bit for bit -- and against the oracle: 9 500 rows = 5 blocks, empty rows,
    rows of more than 64 n-grams, duplicate rows in every block, a row that overflows its push slots."""
    rng = np.random.default_rng(77)
    n, n_col = 9500, 700
    a3 = _sym_lists(rng, n, n_col)
    monkeypatch.setenv("PFZ_K3_SYM", "0")
    ref_idx, ref_val = _self_match(ctx, a3, n_col + 2, ntop, lb)
    if ntop == 5:
        exp_idx, exp_val = oracle_mod.cossim_topn(a3, a3, n_col + 2, ntop, lb, exclude_diag=True)
        assert_topn_parity(ref_idx, ref_val, exp_idx, exp_val, oracle_mod, a3, a3, n_col + 2, exclude_diag=True)
    monkeypatch.setenv("PFZ_K3_SYM", "1")
    census = []
    idx, val = _self_match(ctx, a3, n_col + 2, ntop, lb, repeats=2, census=census)      # twice: the session buffers are re-used
    np.testing.assert_array_equal(idx, ref_idx)
    np.testing.assert_array_equal(val, ref_val)
    magnets, recomputed = census[0]
    assert magnets == 4 * 32                  # one per LDS bank class of the blocks 1 .. 4 (block 0 has nothing below it)
    if lb == 0.0:
        assert recomputed >= 1                # the second weird row overflowed its push slots


def test_symmetric_kernel_in_row_ranges(ctx, monkeypatch):
    """A self-match enqueued in ascending row ranges (what TFIDF.match does so that frame building overlaps the device): the
    symmetric session carries over; ranges that do not continue it fall back to the row-major kernel -- same result always."""
    rng = np.random.default_rng(78)
    n, n_col = 9500, 700
    a3 = _sym_lists(rng, n, n_col)
    monkeypatch.setenv("PFZ_K3_SYM", "0")
    ref_idx, ref_val = _self_match(ctx, a3, n_col + 2, 5, 0.0)
    monkeypatch.setenv("PFZ_K3_SYM", "1")
    for ranges in ([(0, 3000), (3000, 3001), (3001, 8100), (8100, n)],      # a session in four parts
                   [(0, 4096), (4096, n)],                                   # cuts on a block boundary
                   [(0, 2000), (5000, n), (2000, 5000)],                     # the second range does not continue: row-major
                   [(4000, n), (0, 4000)]):                                  # starts in the middle: row-major, then a session's first part
        idx, val = _self_match(ctx, a3, n_col + 2, 5, 0.0, ranges=ranges)
        np.testing.assert_array_equal(idx, ref_idx)
        np.testing.assert_array_equal(val, ref_val)


def test_symmetric_kernel_on_real_names(ctx, oracle_mod, monkeypatch):
    """40 000 real company names against themselves (20 blocks; the automatic choice takes the symmetric kernel from 20 480
    rows on): equal to the row-major kernel bit for bit, top-5 and top-1, and the unnormalised-matrix scale path."""
    from polyfuzz_amd import datasets, _lib
    from polyfuzz_amd.models import TFIDF
    names = datasets.load_company_names()[:40000]
    res = {}
    for sym in ("0", "auto"):
        if sym == "auto":
            monkeypatch.delenv("PFZ_K3_SYM", raising=False)
        else:
            monkeypatch.setenv("PFZ_K3_SYM", sym)
        for ntop in (5, 1):
            m = TFIDF(min_similarity=0.0, top_n=ntop)
            res[sym, ntop] = m.match_device(names).download()
    for ntop in (5, 1):
        np.testing.assert_array_equal(res["auto", ntop][0], res["0", ntop][0])
        np.testing.assert_array_equal(res["auto", ntop][1], res["0", ntop][1])
    # rows with norms far from 1 (the fixed-point scale follows the norm bound): still symmetric bit for bit
    rng = np.random.default_rng(5)
    a3 = _sym_lists(rng, 6000, 500, weird=False)
    a3 = (a3[0], a3[1], a3[2] * 7.5)
    monkeypatch.setenv("PFZ_K3_SYM", "0")
    r0 = _self_match(ctx, a3, 501, 5, 0.0)
    monkeypatch.setenv("PFZ_K3_SYM", "1")
    r1 = _self_match(ctx, a3, 501, 5, 0.0)
    np.testing.assert_array_equal(r1[0], r0[0])
    np.testing.assert_array_equal(r1[1], r0[1])


@pytest.mark.parametrize("n,ntop", [(20480, 5), (30001, 32), (45056, 1)])
def test_streamed_session_at_other_sizes(ctx, monkeypatch, n, ntop):
    """Round 6: `pfz_cossim_topn_ranges` (one pass-1 launch, the ranges merged on a side stream and mirrored into pinned host
    memory) away from the headline's size: the smallest list the symmetric form takes (ten to-blocks), a list whose last block is
    nearly empty, top_n at the form's limit of 32 and at 1, ranges of one block -- against the row-major kernel bit for bit, through
    the host mirror and in the device buffer; and `TFIDF.match` (which enqueues it from 40 000 rows on) against the same matcher
    with the streamed form switched off."""
    import ctypes
    from polyfuzz_amd import _lib, synth
    from polyfuzz_amd.models import TFIDF
    from polyfuzz_amd.models._tfidf import _SPLIT_EVENT
    names = synth.company_names(n, seed=n)
    s = _lib.DeviceStrings.upload(ctx, names)
    a = _lib.DeviceTfidf.fit(ctx, _lib.TfidfParams(3, 3, 1, 1), s, None).transform(s)
    monkeypatch.setenv("PFZ_K3_SYM", "0")
    r_idx, r_val = _lib.cossim_topn(ctx, _lib.DeviceIndex.build(ctx, a), a, ntop, 0.0, exclude_diag=True).download()
    monkeypatch.delenv("PFZ_K3_SYM", raising=False)
    for ends in ([n], [2048 * (i + 1) for i in range(min(15, (n - 1) // 2048))] + [n], [4096, 6144, n]):
        ix = _lib.DeviceIndex.build(ctx, a)
        res, h_idx, h_val = _lib.cossim_topn_ranges(ctx, ix, a, ntop, 0.0, True, ends, _SPLIT_EVENT, mirror=True)
        assert h_idx and ix.symmetric_launches() == (1, n)
        row0 = 0
        for i, row1 in enumerate(ends):
            ctx.event_wait(_SPLIT_EVENT + i)
            m = (row1 - row0) * ntop
            g_idx = np.ctypeslib.as_array(ctypes.cast(h_idx + 4 * ntop * row0, ctypes.POINTER(ctypes.c_int32)), (m,)).reshape(-1, ntop)
            g_val = np.ctypeslib.as_array(ctypes.cast(h_val + 4 * ntop * row0, ctypes.POINTER(ctypes.c_float)), (m,)).reshape(-1, ntop)
            np.testing.assert_array_equal(g_idx, r_idx[row0:row1])
            np.testing.assert_array_equal(g_val, r_val[row0:row1])
            row0 = row1
        d_idx, d_val = res.download()
        np.testing.assert_array_equal(d_idx, r_idx)
        np.testing.assert_array_equal(d_val, r_val)
    if n >= 40000:
        frames = {}
        for knob in ("0", "1"):
            monkeypatch.setenv("PFZ_K3_NO_STREAMED", knob) if knob == "1" else monkeypatch.delenv("PFZ_K3_NO_STREAMED", raising=False)
            frames[knob] = TFIDF(min_similarity=0, top_n=2).match(names)
        assert frames["0"].equals(frames["1"])


@pytest.mark.parametrize("n,expect_sym", [(20479, 0), (20480, 1)])
def test_symmetric_switch_over_small(ctx, monkeypatch, n, expect_sym):
    """Either side of the automatic choice's lower end (20 480 rows = ten to-blocks): one row fewer runs the row-major kernel,
    the size itself the symmetric one -- the same result as the forced row-major run, bit for bit."""
    from polyfuzz_amd import datasets, _lib
    names = datasets.load_company_names()[:n]
    s = _lib.DeviceStrings.upload(ctx, names)
    a = _lib.DeviceTfidf.fit(ctx, _lib.TfidfParams(3, 3, 1, 1), s, None).transform(s)
    monkeypatch.setenv("PFZ_K3_SYM", "0")
    ref = _lib.cossim_topn(ctx, _lib.DeviceIndex.build(ctx, a), a, 5, 0.0, exclude_diag=True).download()
    monkeypatch.delenv("PFZ_K3_SYM", raising=False)
    ix = _lib.DeviceIndex.build(ctx, a)
    got = _lib.cossim_topn(ctx, ix, a, 5, 0.0, exclude_diag=True).download()
    assert ix.symmetric_launches()[0] == expect_sym
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])


@pytest.mark.parametrize("n,expect_sym", [(250000, 1), (250001, 0)])
def test_symmetric_switch_over_large(ctx, monkeypatch, n, expect_sym):
    """Either side of the upper end (250 000 rows: beyond, the lock-step kernel takes the self-match): synthetic
    company-name-like strings, top-3; the forced row-major run is the reference for both."""
    from polyfuzz_amd import synth, _lib
    names = synth.company_names(n, seed=11)
    s = _lib.DeviceStrings.upload(ctx, names)
    a = _lib.DeviceTfidf.fit(ctx, _lib.TfidfParams(3, 3, 1, 1), s, None).transform(s)
    monkeypatch.setenv("PFZ_K3_SYM", "0")
    monkeypatch.setenv("PFZ_K3_LOCKSTEP", "0")
    ref = _lib.cossim_topn(ctx, _lib.DeviceIndex.build(ctx, a), a, 3, 0.0, exclude_diag=True).download()
    monkeypatch.delenv("PFZ_K3_SYM", raising=False)
    monkeypatch.delenv("PFZ_K3_LOCKSTEP", raising=False)
    ix = _lib.DeviceIndex.build(ctx, a)
    got = _lib.cossim_topn(ctx, ix, a, 3, 0.0, exclude_diag=True).download()
    assert ix.symmetric_launches()[0] == expect_sym
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    # at this size the BOUND of the index' pieces (characters / 16 + one per list: ~2.0 M) is beyond the 2^20 up to
    # which the build sizes the postings by the bound, so it waits for the exact count -- which must reach `n_pieces` (it stayed 0:
    # the symmetric form re-dealt the dummy piece only and read garbage postings) and size the postings exactly
    info = ix.info()
    assert info["n_pieces"] * info["piece_postings"] >= info["nnz"] > 0
    assert info["n_pieces"] <= info["nnz"] // info["piece_postings"] + info["n_cols"] * info["n_blocks"]
