"""GPU parity of K5 (dense cosine top-n, fp32 MFMA) against the float64 oracle and against DataFrames the
REFERENCE produced from its own embedding fixtures (tests/golden/dense_golden.*)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _check(idx, val, e_idx, e_val, dense, tol=1e-5):
    np.testing.assert_allclose(val, e_val, rtol=0, atol=tol)
    bad = np.nonzero((idx != e_idx).any(axis=1))[0]
    for i in bad:                       # only float64 near-ties may swap
        for r in range(idx.shape[1]):
            if idx[i, r] != e_idx[i, r]:
                s = dense[i, idx[i, r]] if idx[i, r] >= 0 else 0.0
                assert abs(s - e_val[i, r]) < 4e-6, (i, r, idx[i], e_idx[i])
    assert len(bad) <= max(1, len(idx) // 100)


@pytest.mark.parametrize("n_a,n_b,d,ntop", [(6, 3, 300, 2), (1, 1, 1, 1), (130, 257, 768, 5), (300, 1000, 33, 10),
                                            (513, 129, 64, 128), (70, 2000, 96, 400)])
def test_random_dense_vs_oracle(ctx, oracle_mod, n_a, n_b, d, ntop):
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(n_a + n_b + d)
    a = rng.standard_normal((n_a, d)).astype(np.float32)
    b = rng.standard_normal((n_b, d)).astype(np.float32)
    if n_b > 10:
        b[3] = a[0] * 2.5            # an exact-direction duplicate: cosine 1
        b[7] = 0                     # zero row: cosine 0 with everything
    idx, val = _lib.dense_cossim_topn_host(ctx, a, b, ntop, 0.0)
    e_idx, e_val = oracle_mod.dense_cossim_topn(a, b, ntop, 0.0)
    _check(idx, val, e_idx, e_val, oracle_mod.dense_cossim(a, b))
    if n_b > 10:
        assert idx[0, 0] == 3 and abs(val[0, 0] - 1.0) < 1e-6


@pytest.mark.parametrize("n_a,n_b,ntop,excl", [(70, 3000, 2500, False), (300, 1100, 1025, False), (1500, 1500, 1499, True)])
def test_deep_top_n_in_passes(ctx, oracle_mod, monkeypatch, n_a, n_b, ntop, excl):
    """Round 6 (VERDICT r5 missing 6): top_n beyond the 1024 keys one pass of k5_row_topn keeps -- the reference clips top_n to
    the number of distinct to-strings only (_utils.py:54-56) --: passes of 1024 over the same score panel, each continuing
    strictly below the last key of the one before.  Against the float64 oracle: every score within 1e-5, an index may differ only
    between float64 near-ties; exact duplicates (equal scores) by ascending column across a pass boundary; rows that run out of
    candidates (lower bound) end in -1 / 0; several panels."""
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(n_a + ntop)
    a = rng.standard_normal((n_a, 24)).astype(np.float32)
    b = a.copy() if excl else rng.standard_normal((n_b, 24)).astype(np.float32)
    if not excl:
        b[100:140] = b[50]                         # forty exact duplicates: one score, forty columns in ascending order
    lb = 0.0 if n_b != 1100 else -1.0            # (non-positive scores are "no match": about half of a random row's columns)
    monkeypatch.setenv("PFZ_K5_PANEL_ROWS", "128")
    idx, val = _lib.dense_cossim_topn_host(ctx, a, b, ntop, lb, exclude_diag=excl)
    e_idx, e_val = oracle_mod.dense_cossim_topn(a, b, ntop, lb, exclude_diag=excl, chunk_rows=64)
    dense = oracle_mod.dense_cossim(a, b)
    np.testing.assert_allclose(val, e_val, rtol=0, atol=1e-5)
    assert ((idx < 0) == (e_idx < 0)).all() and (idx < 0).any()          # (rows run out of positive scores before top_n)
    rr, cc = np.nonzero(idx != e_idx)
    assert np.abs(dense[rr, idx[rr, cc]] - e_val[rr, cc]).max(initial=0.0) < 4e-6     # only near-ties swap
    for i in range(0, n_a, 7):                    # no column twice, the diagonal never
        real = idx[i][idx[i] >= 0]
        assert len(set(real.tolist())) == len(real) and (not excl or i not in real.tolist())
    if not excl:
        dup_rows = np.nonzero((idx == 100).any(axis=1))[0]
        assert len(dup_rows) > 0
        for i in dup_rows[:10]:
            at = int(np.nonzero(idx[i] == 50)[0][0])
            np.testing.assert_array_equal(idx[i, at:at + 41], [50] + list(range(100, 140)))


def test_self_match_and_lower_bound(ctx, oracle_mod):
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(9)
    a = rng.standard_normal((400, 96)).astype(np.float32)
    a[100:110] = a[:10] + 0.05 * rng.standard_normal((10, 96)).astype(np.float32)      # near-duplicates
    idx, val = _lib.dense_cossim_topn_host(ctx, a, a, 3, 0.2, exclude_diag=True)
    e_idx, e_val = oracle_mod.dense_cossim_topn(a, a, 3, 0.2, exclude_diag=True)
    _check(idx, val, e_idx, e_val, oracle_mod.dense_cossim(a, a))
    assert (idx[:, 0] != np.arange(400)).all() and (idx[:10, 0] == np.arange(100, 110)).all()


def test_raw_dot_products_of_unnormalised_vectors(ctx, oracle_mod):
    """pfz_dense_dot_topn_host: what the reference's "sparse" back-end forms from dense input
    (_utils.py:74-82) -- no normalisation, so the ranking follows the dot product, not the cosine."""
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(21)
    a = (rng.standard_normal((150, 70)) * rng.uniform(0.2, 3.0, (150, 1))).astype(np.float32)
    b = (rng.standard_normal((900, 70)) * rng.uniform(0.2, 3.0, (900, 1))).astype(np.float32)
    idx, val = _lib.dense_cossim_topn_host(ctx, a, b, 6, 0.5, normalize=False)
    e_idx, e_val = oracle_mod.dense_cossim_topn(a, b, 6, 0.5, normalize=False)
    dots = oracle_mod.dense_cossim(a, b, normalize=False)
    np.testing.assert_allclose(val, e_val, rtol=2e-6, atol=2e-5)
    bad = np.nonzero((idx != e_idx).any(axis=1))[0]
    for i in bad:                                    # only fp32-level near-ties may swap
        for r in range(idx.shape[1]):
            if idx[i, r] != e_idx[i, r]:
                s_got = dots[i, idx[i, r]] if idx[i, r] >= 0 else 0.0
                assert abs(s_got - e_val[i, r]) < 1e-4 * max(1.0, abs(e_val[i, r]))
    assert len(bad) <= 3
    c_idx, _ = _lib.dense_cossim_topn_host(ctx, a, b, 6, 0.0)
    assert (c_idx[:, 0] != idx[:, 0]).any()           # cosine and dot product rank differently here


def test_embeddings_matcher(ctx, golden_dense):
    """polyfuzz_amd.models.Embeddings with ready-made vectors: the reference's frames for its own
    fixtures, `re_train=False` reuse of the stored to-side, a callable embedding_method, loud errors."""
    from polyfuzz_amd.models import Embeddings
    g, cases = golden_dense
    fv, tv = g["from_vec"], g["to_vec"]
    fl, tl = cases["from_list"], cases["to_list"]
    case = next(c for c in cases["cases"] if c["top_n"] == 2 and not c["self"])
    m = Embeddings(min_similarity=0.0, top_n=2, cosine_method="sklearn")
    df = m.match(fl, tl, embeddings_from=fv, embeddings_to=tv)
    assert m.type == "Embeddings" and m.embeddings_to is tv
    for c in df.columns:
        if "Similarity" in c:
            np.testing.assert_allclose(np.array(df[c].tolist(), float), np.array(case["df"][c], float), atol=1.01e-3)
        else:
            assert df[c].tolist() == case["df"][c], c
    df2 = m.match(fl, tl, embeddings_from=fv, re_train=False)       # stored to-side
    assert df2.equals(df)
    # unit-norm rows: the "sparse" back-end (raw dot) gives the same frame as the cosine ones
    df3 = Embeddings(min_similarity=0.0, top_n=2, cosine_method="sparse").match(
        fl, tl, embeddings_from=fv, embeddings_to=tv)
    assert df3["To"].tolist() == df["To"].tolist()
    np.testing.assert_allclose(df3["Similarity"].to_numpy(), df["Similarity"].to_numpy(), atol=1.01e-3)
    # a callable embedding method (bag of characters) and the self-match form
    def bag(strings):
        out = np.zeros((len(strings), 26))
        for i, s in enumerate(strings):
            for ch in s:
                if "a" <= ch <= "z":
                    out[i, ord(ch) - 97] += 1
        return out
    me = Embeddings(bag, min_similarity=0.0, cosine_method="hip")
    d = me.match(["apple", "apples", "house"], ["mouse", "appel"])
    assert d["To"].tolist() == ["appel", "appel", "mouse"]
    ds = me.match(["apple", "apples", "house"])
    assert ds["To"].tolist() == ["apples", "apple", "apples"]
    with pytest.raises(ValueError, match="embedding_method"):
        Embeddings().match(["a"], ["b"])
    with pytest.raises(ValueError, match="no to-side"):
        Embeddings(bag).match(["a"], ["b"], re_train=False)
    with pytest.raises(TypeError):
        Embeddings(embedding_method=[object()])


def test_reference_embedding_fixtures(golden_dense):
    """cosine_similarity operator on the reference's unit-norm 300-d fixtures == the reference's own frames."""
    from polyfuzz_amd.models import cosine_similarity
    g, cases = golden_dense
    for case in cases["cases"]:
        to_list = None if case["self"] else cases["to_list"]
        b = g["from_vec"] if case["self"] else g["to_vec"]
        df = cosine_similarity(g["from_vec"], b, cases["from_list"], to_list, min_similarity=0.0,
                               top_n=case["top_n"], method="sklearn")
        assert list(df.columns) == list(case["df"].keys())
        for c in df.columns:
            got = df[c].tolist()
            if "Similarity" in c:
                np.testing.assert_allclose(np.array(got, float), np.array(case["df"][c], float), atol=1.01e-3)
            else:
                assert got == case["df"][c], c


@pytest.fixture(scope="module")
def golden_dense():
    g = np.load(os.path.join(HERE, "golden", "dense_golden.npz"))
    with open(os.path.join(HERE, "golden", "dense_golden.json")) as f:
        return g, json.load(f)


def test_resident_to_side_and_sharded_dense_job(ctx):
    """The device-resident form of K5: Embeddings keeps the to-vectors in HBM for re_train=False (PolyFuzz.transform),
    and DenseMatchJob shards the from-rows over two contexts on one device (local transport, uneven shards) -- both
    equal the one-shot host entry point bit for bit."""
    import pickle
    import concurrent.futures as cf
    import polyfuzz_amd
    from polyfuzz_amd import _lib, pipeline
    from polyfuzz_amd.models import Embeddings
    rng = np.random.default_rng(21)
    a = rng.standard_normal((301, 96)).astype(np.float32)
    b = rng.standard_normal((530, 96)).astype(np.float32)
    fl, tl = [f"f{i}" for i in range(len(a))], [f"t{i}" for i in range(len(b))]
    ref_idx, ref_val = _lib.dense_cossim_topn_host(ctx, a, b, 4, 0.0)

    m = Embeddings(min_similarity=0.0, top_n=4, cosine_method="hip")
    df = m.match(fl, tl, embeddings_from=a, embeddings_to=b)
    assert df["To"].tolist() == [tl[j] for j in ref_idx[:, 0]]
    df2 = m.match(fl[:50], tl, embeddings_from=a[:50], re_train=False)            # to-side: resident, not re-uploaded
    assert df2["To_3"].tolist() == [tl[j] for j in ref_idx[:50, 2]]
    m2 = pickle.loads(pickle.dumps(m))                                            # device copy left behind, re-created
    assert m2.match(fl[:50], tl, embeddings_from=a[:50], re_train=False).equals(df2)

    ctxs = [polyfuzz_amd.Context(0), polyfuzz_amd.Context(0)]
    comms = _lib.Comm.local_group(ctxs)
    bounds = [pipeline.shard_bounds(len(a), 2, r) for r in range(2)]
    sizes = [e - s for s, e in bounds]

    def rank_fn(r, self_match):
        s, e = bounds[r]
        job = pipeline.DenseMatchJob(ctxs[r], a[s:e], a if self_match else b, top_n=4, comm=comms[r],
                                     rows_per_rank=max(sizes), self_match=self_match, shard_offset=s if self_match else 0)
        idx, val = job.step().download()
        return pipeline.TfidfMatchJob.unpad(idx, val, sizes, max(sizes))

    for self_match in (False, True):
        exp = _lib.dense_cossim_topn_host(ctx, a, a, 4, 0.0, exclude_diag=True) if self_match else (ref_idx, ref_val)
        with cf.ThreadPoolExecutor(2) as ex:
            outs = [f.result(timeout=120) for f in [ex.submit(rank_fn, r, self_match) for r in range(2)]]
        for idx, val in outs:
            np.testing.assert_array_equal(idx, exp[0])
            np.testing.assert_array_equal(val, exp[1])
    for c in comms:
        c.free()


def test_panels_overlap_on_two_streams(ctx, monkeypatch):
    """With more than one score panel the row top-n of panel p runs on a side stream beside the GEMM of panel p + 1
    (two panel buffers, events both ways).  Forcing 128-row panels on a small input gives the one-panel result."""
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(5)
    a = rng.standard_normal((700, 64)).astype(np.float32)
    b = rng.standard_normal((900, 64)).astype(np.float32)
    ref = _lib.dense_cossim_topn_host(ctx, a, b, 6, 0.0)
    monkeypatch.setenv("PFZ_K5_PANEL_ROWS", "128")
    for _ in range(3):                                  # 6 panels each; repeated: buffers and events are reused
        got = _lib.dense_cossim_topn_host(ctx, a, b, 6, 0.0)
        np.testing.assert_array_equal(got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])
    monkeypatch.setenv("PFZ_K5_NO_OVERLAP", "1")
    got = _lib.dense_cossim_topn_host(ctx, a, a, 6, 0.0, exclude_diag=True)
    monkeypatch.delenv("PFZ_K5_NO_OVERLAP")
    got2 = _lib.dense_cossim_topn_host(ctx, a, a, 6, 0.0, exclude_diag=True)
    np.testing.assert_array_equal(got[0], got2[0])
    np.testing.assert_array_equal(got[1], got2[1])


@pytest.mark.parametrize("ntop", [1, 10, 128])
def test_block_maxima_shortcut_equals_full_scan(ctx, oracle_mod, monkeypatch, ntop):
    """The second-generation GEMM leaves per-row maxima of 64-column blocks; the row top-n then reads only the blocks
    whose maximum reaches the ntop-th largest block maximum.  Same result, bit for bit, as scanning every score
    (PFZ_K5_NO_BLOCK_MAX) -- with exact ties from duplicated to-vectors spread over several blocks, a self-match whose
    diagonal (score 1) must not count as a block's winner, a lower bound most rows never reach, rows with fewer than
    ntop matches, edge tiles (n % 128 != 0) and several panels -- and the oracle's result."""
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(77 + ntop)
    d = 64
    b = rng.standard_normal((1500, d)).astype(np.float32)
    for j in (70, 700, 1400, 1499):
        b[j] = b[5]                                         # exact duplicates in four different 64-column blocks
    a = rng.standard_normal((333, d)).astype(np.float32)
    a[:40] = b[5] + 0.3 * rng.standard_normal((40, d)).astype(np.float32)     # rows whose best matches are the duplicates
    for lb, self_match in ((0.0, False), (0.35, False), (0.0, True)):
        x, y = (b, b) if self_match else (a, b)
        monkeypatch.setenv("PFZ_K5_NO_BLOCK_MAX", "1")
        full = _lib.dense_cossim_topn_host(ctx, x, y, ntop, lb, exclude_diag=self_match)
        monkeypatch.delenv("PFZ_K5_NO_BLOCK_MAX")
        fast = _lib.dense_cossim_topn_host(ctx, x, y, ntop, lb, exclude_diag=self_match)
        np.testing.assert_array_equal(fast[0], full[0])
        np.testing.assert_array_equal(fast[1], full[1])
        monkeypatch.setenv("PFZ_K5_PANEL_ROWS", "256")
        paneled = _lib.dense_cossim_topn_host(ctx, x, y, ntop, lb, exclude_diag=self_match)
        monkeypatch.delenv("PFZ_K5_PANEL_ROWS")
        np.testing.assert_array_equal(paneled[0], full[0])
        np.testing.assert_array_equal(paneled[1], full[1])
        e_idx, e_val = oracle_mod.dense_cossim_topn(x, y, ntop, lb, exclude_diag=self_match)
        np.testing.assert_allclose(fast[1], e_val, rtol=0, atol=1e-5)
        if self_match:
            assert (fast[0] != np.arange(len(x))[:, None]).all()
            assert fast[0][5, 0] == 70 and fast[0][70, 0] == 5          # duplicates find each other, lowest index first
