"""RCCL path on one GPU (world = 1): the communicator bootstraps, the sharded fit equals the plain fit,
the result all-gather reproduces the local block, and the pipeline job runs end to end through it.
(N > 1 cannot run on the 1-GPU test box; its protocol is covered by tests/test_sharded_protocol_cpu.py.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_world1_comm_and_sharded_fit(ctx, oracle_mod):
    from polyfuzz_amd import _lib, pipeline, synth
    comm = _lib.Comm.init(ctx, _lib.Comm.unique_id(), 0, 1)
    comm.barrier()
    fl, tl = synth.company_names(700, 3), synth.company_names(900, 4)
    params = _lib.TfidfParams(3, 3, 1, 1)
    f, t = _lib.DeviceStrings.upload(ctx, fl), _lib.DeviceStrings.upload(ctx, tl)
    v1 = _lib.DeviceTfidf.fit(ctx, params, t, f)
    a1 = v1.transform(f).download()
    f2, t2 = _lib.DeviceStrings.upload(ctx, fl), _lib.DeviceStrings.upload(ctx, tl)
    v2 = _lib.tfidf_fit_sharded(ctx, comm, params, t2, f2)
    a2 = v2.transform(f2).download()
    for x, y in zip(a1, a2):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(v1.export()[1], v2.export()[1])

    job = pipeline.TfidfMatchJob(ctx, fl, tl, top_n=4, min_similarity=0.0, comm=comm)
    job.gathered = _lib.DeviceTopN.alloc(ctx, job.n_from, 4)      # force the all-gather leg at world = 1
    out = job.step()
    assert out is job.gathered
    idx, val = out.download()
    l_idx, l_val = job.local.download()
    np.testing.assert_array_equal(idx, l_idx)
    np.testing.assert_array_equal(val, l_val)
    a3, b3, n_col = job.host_matrices()
    e_idx, e_val = oracle_mod.cossim_topn(a3, b3, n_col, 4, 0.0)
    assert np.abs(val - e_val).max() <= 1e-5
    assert (idx != e_idx).any(axis=1).sum() <= 1
    comm.free()


@pytest.mark.parametrize("clean", [True, False])
def test_world2_local_transport_sharded_job(oracle_mod, clean):
    """The sharded library code at world = 2 on ONE device: two contexts, one host thread per rank, the "local"
    transport (host rendezvous + device-to-device copies) behind the same seams RCCL sits behind.
    pfz_tfidf_fit_sharded (bitmap all-gather + OR, df / n_docs all-reduce; with clean=False also the alphabet
    all-gather) + pfz_comm_allgather_topn + TfidfMatchJob(rows_per_rank, unpad) with uneven shards
    (301 = 151 + 150) give the single-context result bit for bit, which is the oracle's."""
    import polyfuzz_amd
    from polyfuzz_amd import _lib, pipeline, synth
    fl, tl = synth.company_names(301, 11), synth.company_names(257, 12)
    if not clean:
        fl = [s.title() + (" é" if i % 7 == 0 else "") for i, s in enumerate(fl)]   # rank-dependent alphabets
    ctxs = [polyfuzz_amd.Context(0), polyfuzz_amd.Context(0)]
    comms = _lib.Comm.local_group(ctxs)
    res = pipeline.run_sharded_job(ctxs, comms, fl, tl, top_n=4, min_similarity=0.0, clean_string=clean)
    single = pipeline.TfidfMatchJob(ctxs[0], fl, tl, top_n=4, min_similarity=0.0, clean_string=clean)
    s_idx, s_val = single.step().download()
    for (idx, val), job in res:                       # every rank holds the whole, un-padded result
        np.testing.assert_array_equal(idx, s_idx)
        np.testing.assert_array_equal(val, s_val)
        np.testing.assert_array_equal(job.vec.export()[1], single.vec.export()[1])      # idf
        assert job.vec.info() == single.vec.info()
    a3, b3, n_col = single.host_matrices()
    e_idx, e_val = oracle_mod.cossim_topn(a3, b3, n_col, 4, 0.0)
    assert np.abs(s_val - e_val).max() <= 1e-5 and (s_idx != e_idx).any(axis=1).sum() <= 1

    # row-sharded self-match: fit on the replicated list alone, diagonal at the shard offset
    res = pipeline.run_sharded_job(ctxs, comms, fl, fl, top_n=3, min_similarity=0.0, clean_string=clean, self_match=True)
    whole = pipeline.TfidfMatchJob(ctxs[1], fl, None, top_n=3, min_similarity=0.0, clean_string=clean, self_match=True)
    w_idx, w_val = whole.step().download()
    for (idx, val), job in res:
        np.testing.assert_array_equal(idx, w_idx)
        np.testing.assert_array_equal(val, w_val)
        assert job.vec.info()["n_docs"] == len(fl)
    del res
    for c in comms:
        c.free()


def test_world2_sharded_fit_of_wide_ngram_codes(oracle_mod):
    """n_gram_range=(3, 6) -- the range the reference's own test uses (tests/test_polyfuzz.py:111) -- gives 36-bit codes
    of cleaned text: the sorted-vocabulary path.  Its sharded fit (all-gather of the ranks' distinct codes, one sort,
    distinct again) must build the vectoriser of the single-context fit; so must (6, 10): 60-bit codes."""
    import polyfuzz_amd
    from polyfuzz_amd import _lib, pipeline, synth
    fl, tl = synth.company_names(203, 31), synth.company_names(180, 32)
    ctxs = [polyfuzz_amd.Context(0), polyfuzz_amd.Context(0)]
    comms = _lib.Comm.local_group(ctxs)
    for rng in ((3, 6), (6, 10)):
        res = pipeline.run_sharded_job(ctxs, comms, fl, tl, top_n=3, min_similarity=0.0, n_gram_range=rng)
        single = pipeline.TfidfMatchJob(ctxs[0], fl, tl, top_n=3, min_similarity=0.0, n_gram_range=rng)
        s_idx, s_val = single.step().download()
        assert single.vec.info()["code_bits"] > 32
        for (idx, val), job in res:
            assert job.vec.info() == single.vec.info()
            np.testing.assert_array_equal(job.vec.export()[0], single.vec.export()[0])      # the n-grams themselves
            np.testing.assert_array_equal(job.vec.export()[1], single.vec.export()[1])      # idf
            np.testing.assert_array_equal(idx, s_idx)
            np.testing.assert_array_equal(val, s_val)
        a3, b3, n_col = single.host_matrices()
        e_idx, e_val = oracle_mod.cossim_topn(a3, b3, n_col, 3, 0.0)
        assert np.abs(s_val - e_val).max() <= 1e-5
    del res
    for c in comms:
        c.free()


@pytest.mark.parametrize("self_match", [False, True])
def test_world2_to_side_sharding_and_merge(oracle_mod, self_match):
    """The north-star's variant: the TO-list sharded over two ranks (uneven: 257 = 129 + 128), every rank matches all
    from-rows against its shard, candidates all-gathered and merged (pfz_comm_merge_to_shards).  Same result as the
    one-context match up to near-ties (each shard has its own fixed-point scale), which is the oracle's."""
    import concurrent.futures as cf
    import polyfuzz_amd
    from polyfuzz_amd import _lib, pipeline, synth
    fl, tl = synth.company_names(301, 11), synth.company_names(257, 12)
    if self_match:
        tl = fl
    ctxs = [polyfuzz_amd.Context(0), polyfuzz_amd.Context(0)]
    comms = _lib.Comm.local_group(ctxs)

    def rank_fn(r):
        b, e = pipeline.shard_bounds(len(tl), 2, r)
        job = pipeline.ToShardedMatchJob(ctxs[r], fl, tl[b:e], b, comms[r], top_n=4, min_similarity=0.0, self_match=self_match)
        out = job.step().download()
        return out, job.vec.export()[1]

    with cf.ThreadPoolExecutor(2) as ex:
        outs = [f.result(timeout=120) for f in [ex.submit(rank_fn, r) for r in range(2)]]
    single = pipeline.TfidfMatchJob(ctxs[0], fl, None if self_match else tl, top_n=4, min_similarity=0.0, self_match=self_match)
    s_idx, s_val = single.step().download()
    a3, b3, n_col = single.host_matrices()
    e_idx, e_val = oracle_mod.cossim_topn(a3, b3, n_col, 4, 0.0, exclude_diag=self_match)
    for (idx, val), idf in outs:
        np.testing.assert_array_equal(idf, single.vec.export()[1])          # the sharded fit is the global fit
        assert np.abs(val - e_val).max() <= 1e-5
        assert (idx != e_idx).any(axis=1).sum() <= 2
        # bit-identical to the single-context job: device-vectorised rows are L2-normalised, every shard's index and the
        # single index use the same fixed-point scale (norm bound 1), and integer sums do not depend on how the to-rows
        # were cut -- the merge is by (sum desc, GLOBAL index asc), the single kernel's own order
        np.testing.assert_array_equal(idx, s_idx)
        np.testing.assert_array_equal(val, s_val)
    np.testing.assert_array_equal(outs[0][0][0], outs[1][0][0])
    for c in comms:
        c.free()


def test_world2_best_choice_job_edit_distance_and_wratio(oracle_mod):
    """SURVEY section 8e for the edit-distance matchers: from-strings sharded over two ranks (uneven: 61 = 31 + 30), the
    to-list replicated, (index, float64 score) blocks all-gathered through the library's transport -- the exact
    scores arrive on every rank (K4 ratio and K7 WRatio), equal to the single-context result and to the oracles."""
    import concurrent.futures as cf
    import polyfuzz_amd
    from oracle import fuzz_scorers as f
    from polyfuzz_amd import _lib, pipeline, synth
    fl, tl = synth.company_names(61, 21), synth.company_names(150, 22)
    ctxs = [polyfuzz_amd.Context(0), polyfuzz_amd.Context(0)]
    comms = _lib.Comm.local_group(ctxs)
    bounds = [pipeline.shard_bounds(len(fl), 2, r) for r in range(2)]
    sizes = [e - b for b, e in bounds]
    rpr = max(sizes)
    fl = list(fl)
    fl[3] = fl[40] = fl[60] = ""          # empty from-strings in BOTH shards: QRatio scores them 0 against everything
    tl = list(tl)
    tl[7] = ""
    for scorer in ("ratio", "WRatio", "QRatio"):
        def rank_fn(r):
            b, e = bounds[r]
            job = pipeline.BestChoiceJob(ctxs[r], fl[b:e], tl, scorer=scorer, comm=comms[r], rows_per_rank=rpr)
            return pipeline.BestChoiceJob.unpad(*job.result_host(job.step()), sizes, rpr)
        with cf.ThreadPoolExecutor(2) as ex:
            outs = [x.result(timeout=120) for x in [ex.submit(rank_fn, r) for r in range(2)]]
        job1 = pipeline.BestChoiceJob(ctxs[0], fl, tl, scorer=scorer)
        single = job1.result_host(job1.step())
        if scorer == "ratio":
            e_idx, e_score = oracle_mod.indel_argmax(fl, tl)
        else:
            e_idx, e_score = f.extract_one_all(fl, tl, f.SCORERS[scorer])
        for idx, score in outs:
            np.testing.assert_array_equal(idx, single[0])
            np.testing.assert_array_equal(score, single[1])
            np.testing.assert_array_equal(idx, np.array(e_idx, np.int32))
            np.testing.assert_array_equal(score, np.array(e_score, np.float64))
    for c in comms:
        c.free()


@pytest.mark.parametrize("world,top_n", [(2, 5), (3, 1), (4, 8)])
def test_symmetric_self_match_cut_over_local_ranks(monkeypatch, world, top_n):
    """the headline's symmetric form (every unordered pair of rows scored once) survives at N > 1.  `world`
    contexts on ONE device, one host thread per rank, the local transport: rank r works on the rows r, r + world, ... of the
    replicated list, the ranks' pass-0 thresholds and their per-row candidate lists are all-gathered
    (pfz_comm_cossim_topn_symmetric), every rank ends with the FULL result -- the single-context result bit for bit (which
    is the row-major kernel's: tests/test_k3_cossim_gpu.py).  12 000 real names = 6 blocks, the last one partial; odd sizes so
    that the parts' threshold stretches are uneven."""
    import polyfuzz_amd
    from polyfuzz_amd import _lib, datasets, pipeline
    names = datasets.load_company_names()[:12001]
    monkeypatch.setenv("PFZ_K3_SYM", "1")               # (the automatic choice starts at 20 480 rows)
    ctxs = [polyfuzz_amd.Context(0) for _ in range(world)]
    comms = _lib.Comm.local_group(ctxs)
    res = pipeline.run_sharded_job(ctxs, comms, names, names, top_n=top_n, min_similarity=0.0, self_match=True)
    monkeypatch.setenv("PFZ_K3_SYM", "0")
    whole = pipeline.TfidfMatchJob(ctxs[0], names, None, top_n=top_n, min_similarity=0.0, self_match=True)
    w_idx, w_val = whole.step().download()
    assert whole.index.symmetric_launches()[0] == 0     # the reference run is the row-major kernel's
    for (idx, val), job in res:
        assert job.result_is_full and job.index.symmetric_launches() == (1, len(names))
        np.testing.assert_array_equal(idx, w_idx)
        np.testing.assert_array_equal(val, w_val)
    # a second step on the same jobs (buffers re-used) and the row-sharded form of the same job (PFZ_K3_SYM=0) agree too
    res0 = pipeline.run_sharded_job(ctxs, comms, names, names, top_n=top_n, min_similarity=0.0, self_match=True)
    for (idx, val), job in res0:
        assert not job.result_is_full
        np.testing.assert_array_equal(idx, w_idx)
        np.testing.assert_array_equal(val, w_val)
    del res, res0
    for c in comms:
        c.free()
