"""world_size-2 (gloo, CPU) check of the multi-GPU protocol of polyfuzz_amd.pipeline /
pfz_tfidf_fit_sharded: row-sharded from-list, replicated to-list,
  vocabulary = union of per-rank n-gram sets (device: all-gather of bitmaps + OR),
  df / n_docs = sum over ranks with the replicated list counted by rank 0 only (device: all-reduce),
  result = concatenation of the per-shard top-n blocks (device: all-gather).
The exchange runs here over torch.distributed/gloo with the oracle doing the per-rank arithmetic; the
assertion is that the sharded protocol reproduces the single-process fit + match exactly."""
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
    import oracle
    from polyfuzz_amd.pipeline import shard_bounds
    from_list, to_list = _lists()
    b, e = shard_bounds(len(from_list), WORLD, rank)
    shard = from_list[b:e]

    # --- sharded fit -------------------------------------------------------------
    counted = (to_list if rank == 0 else []) + shard           # replicated list counted once
    local_df = {}
    for s in counted:
        for g in set(oracle.create_ngrams(s)):
            local_df[g] = local_df.get(g, 0) + 1
    local_vocab = set(local_df) | {g for s in to_list for g in oracle.create_ngrams(s)}
    gathered = [None] * WORLD
    dist.all_gather_object(gathered, sorted(local_vocab))        # device: all-gather of code bitmaps
    vocab = sorted(set().union(*gathered))                       # device: OR + rank prefix
    df = torch.tensor([local_df.get(g, 0) for g in vocab], dtype=torch.int64)
    n_docs = torch.tensor([len(counted)], dtype=torch.int64)
    dist.all_reduce(df)                                          # device: ncclAllReduce(sum) of df
    dist.all_reduce(n_docs)
    v = oracle.TfidfOracle()
    v.vocabulary = vocab
    v.index = {g: i for i, g in enumerate(vocab)}
    v.df = df.numpy()
    v.n_docs = int(n_docs.item())
    v.idf = np.log((v.n_docs + 1.0) / (v.df.astype(np.float64) + 1.0)) + 1.0

    # --- per-shard match, padded all-gather of the top-n blocks -------------------
    a3, b3 = v.transform(shard), v.transform(to_list)
    idx, val = oracle.cossim_topn(a3, b3, len(vocab), TOP_N, 0.0)
    rows = max(shard_bounds(len(from_list), WORLD, r)[1] - shard_bounds(len(from_list), WORLD, r)[0]
               for r in range(WORLD))
    pad_idx = torch.full((rows, TOP_N), -1, dtype=torch.int32)
    pad_val = torch.zeros((rows, TOP_N), dtype=torch.float64)
    pad_idx[:len(idx)] = torch.from_numpy(idx)
    pad_val[:len(val)] = torch.from_numpy(val)
    all_idx = [torch.empty_like(pad_idx) for _ in range(WORLD)]
    all_val = [torch.empty_like(pad_val) for _ in range(WORLD)]
    dist.all_gather(all_idx, pad_idx)
    dist.all_gather(all_val, pad_val)
    if rank == 0:
        full_idx, full_val = [], []
        for r in range(WORLD):
            rb, re_ = shard_bounds(len(from_list), WORLD, r)
            full_idx.append(all_idx[r][:re_ - rb].numpy())
            full_val.append(all_val[r][:re_ - rb].numpy())
        np.savez(os.path.join(out_dir, "sharded.npz"), idx=np.concatenate(full_idx), val=np.concatenate(full_val),
                 idf=v.idf, vocab=np.array(vocab))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_protocol_reproduces_single_process_result(tmp_path, oracle_mod):
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    got = np.load(os.path.join(str(tmp_path), "sharded.npz"))
    from_list, to_list = _lists()
    v = oracle_mod.TfidfOracle().fit(to_list + from_list)        # reference _tfidf.py:109
    assert list(got["vocab"]) == v.vocabulary
    np.testing.assert_array_equal(got["idf"], v.idf)
    e_idx, e_val = oracle_mod.cossim_topn(v.transform(from_list), v.transform(to_list), len(v.vocabulary), TOP_N, 0.0)
    np.testing.assert_array_equal(got["idx"], e_idx)
    np.testing.assert_array_equal(got["val"], e_val)
