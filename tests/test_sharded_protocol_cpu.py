"""world_size-2 (gloo, CPU) run of the row-sharded match job -- the code of polyfuzz_amd/pipeline.py itself:
`TfidfMatchJob` (which rank fits on what, the padded result blocks, diagonal offsets of a self-match shard,
the order of the exchanges, `unpad`) driven through the engine seam by tests/cpu_engine.py (oracle arithmetic,
exchanges over torch.distributed/gloo, mirroring pfz_tfidf_fit_sharded and pfz_comm_allgather_topn).  The
assertion is that the sharded job reproduces the single-process fit + match exactly, with uneven shards.
(The same job on the real engine at world = 2: tests/test_comm_gpu.py.)"""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist          # noqa: E402
import torch.multiprocessing as mp        # noqa: E402

WORLD = 2
TOP_N = 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _lists():
    from polyfuzz_amd import synth
    return synth.company_names(301, seed=11), synth.company_names(257, seed=12)   # odd sizes: uneven shards


def _worker(rank, port, out_dir):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from polyfuzz_amd.pipeline import TfidfMatchJob, shard_bounds
    from tests.cpu_engine import GlooComm, OracleEngine
    from_list, to_list = _lists()
    comm, eng = GlooComm(dist), OracleEngine(torch)
    sizes = [shard_bounds(len(from_list), WORLD, r)[1] - shard_bounds(len(from_list), WORLD, r)[0] for r in range(WORLD)]
    b, e = shard_bounds(len(from_list), WORLD, rank)

    # two lists: exact sharded fit + padded all-gather
    job = TfidfMatchJob(None, from_list[b:e], to_list, top_n=TOP_N, comm=comm, rows_per_rank=max(sizes), engine=eng)
    idx, val = job.step().download()
    idx, val = TfidfMatchJob.unpad(idx, val, sizes, max(sizes))
    # self-match of from_list, row-sharded: fit on the replicated list alone, diagonal at the shard offset
    sjob = TfidfMatchJob(None, from_list[b:e], from_list, top_n=TOP_N, comm=comm, rows_per_rank=max(sizes), engine=eng,
                         self_match=True, shard_offset=b)
    s_idx, s_val = sjob.step().download()
    assert sjob.result_is_full          # (the self-match is cut over the ranks in the symmetric form: full result on every rank)
    s_idx, s_val = sjob.whole_result(s_idx, s_val, sizes)
    # ... and the row-sharded form of the same self-match (what runs where the symmetric form does not apply).  ONE rank alone
    # says no -- a different environment, a failed allocation --: the question is a collective, so BOTH ranks take the
    # row-sharded form instead of issuing different exchanges and waiting for each other
    eng.force_row_shards_on_rank = 1
    rjob = TfidfMatchJob(None, from_list[b:e], from_list, top_n=TOP_N, comm=comm, rows_per_rank=max(sizes), engine=eng,
                         self_match=True, shard_offset=b)
    r_idx, r_val = rjob.step().download()
    assert not rjob.result_is_full
    r_idx, r_val = rjob.whole_result(r_idx, r_val, sizes)
    assert np.array_equal(r_idx, s_idx) and np.array_equal(r_val, s_val)
    # the user-level call on the communicator: every rank gets the full frame (bench.py's timed step at N > 1)
    eng.force_row_shards_on_rank = None
    from polyfuzz_amd import pipeline
    frame = pipeline.sharded_self_match(None, comm, from_list, top_n=TOP_N, min_similarity=0.0, engine=eng)
    assert frame["From"].tolist() == from_list and len(frame.columns) == 1 + 2 * TOP_N
    for r in range(TOP_N):
        exp3 = np.round(s_val[:, r].astype(np.float32).astype(np.float64), 3)
        gone = (exp3 < 0.001) | (s_idx[:, r] < 0)
        assert frame["To" if r == 0 else f"To_{r + 1}"].tolist() == [None if g else from_list[j] for g, j in zip(gone, s_idx[:, r])]
        np.testing.assert_array_equal(frame["Similarity" if r == 0 else f"Similarity_{r + 1}"].to_numpy(), np.where(gone, 0.0, exp3))
    np.savez(os.path.join(out_dir, f"sharded{rank}.npz"), idx=idx, val=val, idf=job.vec.v.idf,
             vocab=np.array(job.vec.v.vocabulary), s_idx=s_idx, s_val=s_val, s_ndocs=sjob.vec.v.n_docs)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_protocol_reproduces_single_process_result(tmp_path, oracle_mod):
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    from_list, to_list = _lists()
    v = oracle_mod.TfidfOracle().fit(to_list + from_list)        # reference _tfidf.py:109
    e_idx, e_val = oracle_mod.cossim_topn(v.transform(from_list), v.transform(to_list), len(v.vocabulary), TOP_N, 0.0)
    vs = oracle_mod.TfidfOracle().fit(from_list)                 # reference _tfidf.py:113-116
    a3 = vs.transform(from_list)
    es_idx, es_val = oracle_mod.cossim_topn(a3, a3, len(vs.vocabulary), TOP_N, 0.0, exclude_diag=True)
    for rank in range(WORLD):                                    # every rank holds the full result
        got = np.load(os.path.join(str(tmp_path), f"sharded{rank}.npz"))
        assert list(got["vocab"]) == v.vocabulary
        np.testing.assert_array_equal(got["idf"], v.idf)
        np.testing.assert_array_equal(got["idx"], e_idx)
        np.testing.assert_array_equal(got["val"], e_val)
        assert int(got["s_ndocs"]) == len(from_list)
        np.testing.assert_array_equal(got["s_idx"], es_idx)
        np.testing.assert_array_equal(got["s_val"], es_val)
