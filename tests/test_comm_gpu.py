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
