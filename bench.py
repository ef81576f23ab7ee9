#!/usr/bin/env python3
"""bench.py -- string-pairs/sec of the TF-IDF cosine top-n hot path on MI355X.

Workload (BASELINE.json metric: "TF-IDF cosine 100k x 100k"): char-3-gram TF-IDF,
cosine top-5, min_similarity 0, 100 000 synthetic company-name-like from-strings
against 100 000 to-strings per GPU (polyfuzz_amd.synth: token recombination of
the real company-name statistics; the reference's data files are HTTP downloads
and are not on the GPU box).  One "step" = one pass of the hot path with the
string lists already resident in HBM: vectorise both lists, build the to-side
inverted index, run the fused cosine top-n kernel (everything `TFIDF.match`
does between receiving the lists and assembling the DataFrame).

Multi-GPU (--gpus N, launched by torch.distributed.run, one process per GPU):
the from-side is row-sharded -- every rank owns its own 100k from-rows -- and the
to-side is replicated ("weak" scaling; the shards are independent, the only
exchange is the RCCL all-gather of the per-shard top-n results, which is part of
the timed step).  torch is used for rendezvous/barrier only, never in the data path.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

N_FROM = 100_000
N_TO = 100_000
TOP_N = 5
MIN_SIM = 0.0
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-from", type=int, default=N_FROM)
    ap.add_argument("--n-to", type=int, default=N_TO)
    ap.add_argument("--top-n", type=int, default=TOP_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import polyfuzz_amd
    from polyfuzz_amd import engine, synth

    ctx = polyfuzz_amd.Context(local_rank)
    info = ctx.info()

    # ---- inputs: replicated to-list, per-rank from-shard -----------------------
    to_list = synth.company_names(args.n_to, seed=5678)
    from_list = synth.company_names(args.n_from, seed=1234 + rank)

    job = engine.TfidfMatchJob(ctx, from_list, to_list, top_n=args.top_n, min_similarity=MIN_SIM,
                               world=world, rank=rank, dist=dist)

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        job.step()
    barrier()
    ctx.prof_enable(True)
    ctx.prof_reset()
    t0 = time.perf_counter()
    ctx.event_record(0)
    for _ in range(args.steps):
        job.step()
    ctx.event_record(1)
    barrier()
    t1 = time.perf_counter()
    ctx.prof_enable(False)
    wall = t1 - t0
    if dist is not None:
        import torch
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    stats = job.stats()                      # nnz, madds, vocab ...
    k3_ms, k3_launches = ctx.prof_get("k3_cossim_topn")
    kernel_ms = {name: ctx.prof_get(name)[0] / max(1, args.steps) for name in engine.PROFILED_KERNELS}
    gpu_ms = ctx.event_elapsed_ms(0, 1)

    pairs_per_step = float(args.n_from) * float(args.n_to) * world
    value = pairs_per_step * args.steps / wall

    out = None
    if rank == 0:
        k3_avg_s = (k3_ms / max(1, k3_launches)) * 1e-3
        bytes_alg = 8.0 * stats["madds"] + 8.0 * stats["nnz_from"] + 8.0 * args.n_from * args.top_n
        achieved = bytes_alg / k3_avg_s / 1e9 if k3_avg_s > 0 else 0.0
        out = {
            "metric": "string-pairs/sec, TF-IDF cosine top-n",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"TF-IDF char-3-gram cosine top-{args.top_n}, min_similarity {MIN_SIM}, "
                            f"{args.n_from}x{args.n_to} synthetic company-name-like strings per GPU "
                            f"(from-side row-sharded, to-side replicated)",
                "n_from_per_gpu": args.n_from, "n_to": args.n_to, "top_n": args.top_n,
                "vocab": stats["vocab"], "nnz_from": stats["nnz_from"], "nnz_to": stats["nnz_to"],
                "multiply_adds": stats["madds"],
                "step": job.step_description(),
                "parallelism": f"row-shard x{world}",
                "device": info["name"],
            },
            "top1_latency_ms": wall / args.steps * 1e3,
            "gpu_ms_per_step_rank0": gpu_ms / args.steps,
            "kernel_ms_per_step": kernel_ms,
            "roofline": {
                "kernel": "k3_cossim_topn",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_bytes_per_launch": bytes_alg,
                "avg_launch_ms": k3_avg_s * 1e3,
                "launches": k3_launches,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = engine.cpu_baseline(job, seconds=args.cpu_seconds)
        # parity spot check of the last step's result against the oracle (not timed)
        out["parity_check"] = engine.spot_check(job)
    barrier()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
