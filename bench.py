#!/usr/bin/env python3
"""bench.py -- string-pairs/sec + top-1 match latency of the TF-IDF cosine top-n hot path on MI355X.

Workload (BASELINE.json metric "TF-IDF cosine 100k x 100k"; SURVEY.md §8d "Headline"): the 100 000 real
SEC-EDGAR company names of the reference's data/company_names.json (shipped gzipped in polyfuzz_amd/data/,
the GPU box has no network), SELF-MATCH, char-3-gram TF-IDF, cosine top-5, min_similarity 0 -- i.e.
`TFIDF(min_similarity=0, top_n=5).match(names)` (reference docs/tutorial/datasets/datasets.md:36-41).

One "step" = one pass of the hot path with the string list already resident in HBM: fit vocabulary + idf on
the list, vectorise it, build the inverted index, run the fused cosine top-n with the diagonal excluded --
everything `TFIDF.match` does between receiving the list and assembling the DataFrame (reference
_tfidf.py:93-98,113-116, _utils.py:54-102).  `value` = N_from * N_to * steps / wall.

The same JSON line also carries, measured after the timed region on rank 0 at N = 1:
  match_wall_ms / match_pairs_per_s  `TFIDF(...).match(names)` from a Python list to the DataFrame
                                     (packing, H2D, device step, D2H, frame) -- what a user sees
  latency.top1_single_query_ms       one query string against the fitted 100k list, top-1, re_train=False
  roofline, cpu_baseline (+ cpu_baseline_arms), parity_check

Multi-GPU (--gpus N, launched by torch.distributed.run, one process per GPU):
  --scaling weak   (default) every rank matches its own 100k from-rows against the replicated real list:
                   the global from-list is N x 100k names (rank 0: the real names = the headline self-match,
                   rank r > 0: synthetic names of the same token statistics); fit on the replicated list,
                   no data-path collective except the all-gather of the per-shard top-n blocks (RCCL)
  --scaling strong the one 100k x 100k self-match, its from-rows split over the N ranks
torch is used for rendezvous / barrier / the max-over-ranks only, never in the data path.

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

N_NAMES = 100_000
TOP_N = 5
MIN_SIM = 0.0
# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # HBM3E 8.0 TB/s spec
LDS_BYTES_PER_CLK_CU = 128.0     # LDS bandwidth per CU
N_CU, CLK_HZ = 256, 2.4e9
LDS_ATOMIC_LANES_PER_S = 4.0e12  # measured ds_add_u32 rate, all CUs (tools/ubench/lds_atomic.hip: 6.6 lanes/clk/CU)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=N_NAMES,
                    help="list length: <= 100000 takes the first n real names, more takes synthetic names")
    ap.add_argument("--top-n", type=int, default=TOP_N)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-match-wall", action="store_true", help="skip the .match() wall time and the latency leg")
    ap.add_argument("--config", choices=("tfidf", "editdistance", "dense"), default="tfidf",
                    help="editdistance: BASELINE.json config 3 (EditDistance, 20k x 20k IMDB titles), single GPU; "
                         "dense: one GPU's shard of config 5 (62.5k x 500k x 768 embeddings, cosine top-10)")
    return ap.parse_args()


def recorded_traffic(args):
    """Bytes per K3 launch that missed L2 (rocprofv3 PMC passes; a bench run cannot collect PMC itself); only
    valid for the workload it was recorded on."""
    try:
        with open(os.path.join(REPO, "profiles", "k3_hbm_traffic.json")) as f:
            rec = json.load(f)
        if rec.get("workload") != workload_key(args):
            return None, None
        return float(rec["hbm_bytes_per_launch"]), rec.get("note")
    except (OSError, KeyError, ValueError):
        return None, None


def workload_key(args):
    return f"company_names[:{args.n}] self-match top-{args.top_n}"


def the_list(args):
    from polyfuzz_amd import datasets, synth
    if args.n <= N_NAMES:
        return datasets.load_company_names()[:args.n], "real"
    return synth.company_names(args.n, seed=5678), "synthetic"


# ---- CPU baselines + parity check (rank 0, N = 1, outside the timed region) -------------------------------

def cpu_baselines_and_check(job, idx, val, seconds, names):
    """Three CPU arms on bounded samples of the same workload, and the parity spot check:
    (ii) oracle/cossim_topn.c on ONE core -- how PolyFuzz calls sparse_dot_topn (_utils.py:82) -- primary;
    (iii) the same on all host cores (row ranges on threads; ctypes releases the GIL);
    (i) the reference's own executable back-end, restated (oracle/reference_path.py: sklearn vectoriser +
        dense cosine + full sorts + frame) at the C2 size, where its dense matrix fits."""
    import concurrent.futures as cf
    import oracle
    from oracle.reference_path import sklearn_backend_match
    oracle.build_native()
    a3, b3, n_col = job.host_matrices()
    n_from = len(a3[0]) - 1
    excl = job.self_match

    def run(r0, r1):
        return oracle.cossim_topn(a3, b3, n_col, job.top_n, job.min_similarity, exclude_diag=excl, rows=(r0, r1))

    probe = min(200, n_from)
    t0 = time.perf_counter()
    run(0, probe)
    per_row = (time.perf_counter() - t0) / max(probe, 1)
    rows = int(max(probe, min(n_from, seconds / max(per_row, 1e-9))))
    t0 = time.perf_counter()
    e_idx, e_val = run(0, rows)
    dt = time.perf_counter() - t0
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    base = {"value": rows * float(job.n_to) / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": f"first {rows} of {n_from} from-rows x all {job.n_to} to-rows, oracle/cossim_topn.c "
                      f"(Gustavson + strict bound + top-{job.top_n}, float64), {dt:.1f} s on 1 of {cores} host "
                      "cores; vectorisation not included"}
    arms = [dict(base, arm="ii: sparse product as PolyFuzz calls it (single thread)")]

    # (iii) all cores: one row range per thread, about half the budget of wall time
    per_thread = int(max(50, min(n_from // max(cores, 1), 0.5 * seconds / max(per_row, 1e-9))))
    ranges = [(t * per_thread, (t + 1) * per_thread) for t in range(cores) if (t + 1) * per_thread <= n_from]
    if ranges:
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(len(ranges)) as ex:
            list(ex.map(lambda r: run(*r), ranges))
        dt3 = time.perf_counter() - t0
        arms.append({"arm": "iii: the same on all host cores", "value": len(ranges) * per_thread * float(job.n_to) / dt3,
                     "unit": "pairs/s", "cores": len(ranges), "kind": "port",
                     "sample": f"{len(ranges)} threads x {per_thread} from-rows x all {job.n_to} to-rows, {dt3:.1f} s"})

    # (i) the reference's executable path at C2 (10k x 10k of the same names; the dense matrix is 800 MB)
    try:
        from polyfuzz_amd import datasets
        n2 = 10_000 if len(names) >= 20_000 else max(100, len(names) // 2)
        fl, tl = datasets.c2_lists(n2) if len(names) == N_NAMES else (names[:n2], names[n2:2 * n2])
        tm = {}
        sklearn_backend_match(fl, tl, top_n=job.top_n, timings=tm)
        arms.append({"arm": "i: the reference's sklearn back-end, restated (vectorise + dense cosine + full sorts + frame)",
                     "value": float(n2) * float(n2) / tm["total_s"], "unit": "pairs/s", "cores": "BLAS threads",
                     "kind": "port", "sample": f"config 2: {n2} x {n2} company names, top-{job.top_n}, end to end "
                     f"{tm['total_s']:.1f} s (vectorise {tm['vectorise_s']:.1f}, cosine+sort {tm['cosine_sort_s']:.1f}, "
                     f"frame {tm['frame_s']:.1f})"})
    except Exception as e:      # a CPU arm must never take the bench line down
        arms.append({"arm": "i: reference sklearn back-end", "error": f"{type(e).__name__}: {e}"})

    g_idx, g_val = idx[:rows], val[:rows].astype(np.float64)
    max_err = float(np.abs(g_val - e_val).max()) if rows else 0.0
    mism = np.nonzero((g_idx != e_idx).any(axis=1))[0]
    ties = int((np.diff(e_val, axis=1) == 0).any(axis=1).sum()) if job.top_n > 1 else 0
    # an index mismatch is a real error unless the float64 oracle itself has the two scores within 2e-6
    hard = 0
    for i in mism[:2000]:
        dense = oracle.cossim_dense(a3, b3, n_col, rows=(int(i), int(i) + 1))[0]
        for r in range(job.top_n):
            if g_idx[i, r] != e_idx[i, r]:
                s = dense[g_idx[i, r]] if g_idx[i, r] >= 0 else 0.0
                if abs(s - e_val[i, r]) >= 2e-6:
                    hard += 1
    check = {"rows_checked": rows, "max_abs_score_err": max_err, "rows_with_index_diff": int(len(mism)),
             "index_diffs_not_near_ties": hard, "rows_with_exact_ties_in_top_n": ties,
             "ok": bool(max_err <= 1e-5 and hard == 0)}
    return base, arms, check


# ---- what a user sees: .match() wall time and single-query latency ----------------------------------------

def match_wall(names, top_n, result_idx, reps=7):
    from polyfuzz_amd.models import TFIDF
    m = TFIDF(n_gram_range=(3, 3), min_similarity=MIN_SIM, top_n=top_n)
    for _ in range(2):
        df = m.match(names)
    ts, stages = [], []
    for _ in range(reps):
        df = None                  # (dropping the previous 100k x 11 frame is ~1 ms of reference counting: not part of a call)
        t0 = time.perf_counter()
        df = m.match(names)
        ts.append((time.perf_counter() - t0) * 1e3)
        stages.append(m.last_timings)
    order = np.argsort(ts)
    med = int(order[len(ts) // 2])
    # the frame is the device result: To == names[idx] wherever the rounded score survives
    same = True
    if result_idx is not None:
        to0 = df["To"].tolist()
        same = all(t is None or t == names[j] for t, j in zip(to0[:5000], result_idx[:5000, 0].tolist()))
    n = float(len(names))
    out = {"match_wall_ms": ts[med], "match_wall_ms_min": min(ts), "match_pairs_per_s": n * n / (ts[med] * 1e-3),
           "match_stages_ms": {k: round(v, 3) for k, v in stages[med].items()},
           "match_what": f"TFIDF(min_similarity=0, top_n={top_n}).match(names): Python list in, DataFrame out "
                         f"(pack, H2D, device step, D2H, frame), median of {reps}",
           "match_frame_consistent_with_device_result": bool(same)}
    return out, m


def top1_latency(m_fit, names, reps=50):
    """The other half of BASELINE.json's metric: top-1 match latency.  One query string against the fitted
    100k list through the drop-in matcher (host str in, DataFrame out: upload, vectorise, K3, download, frame)."""
    from polyfuzz_amd.models import TFIDF
    m = TFIDF(n_gram_range=(3, 3), min_similarity=MIN_SIM, top_n=1)
    m.match(names[:1000], names)                 # fit: vocabulary, idf and the to-side index stay in HBM
    q = [names[len(names) // 2]]
    for _ in range(5):
        m.match(q, names, re_train=False)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        m.match(q, names, re_train=False)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    t0 = time.perf_counter()
    TFIDF(n_gram_range=(3, 3), min_similarity=MIN_SIM, top_n=1).match(names)
    full = (time.perf_counter() - t0) * 1e3
    return {"top1_single_query_ms": ts[len(ts) // 2], "top1_single_query_ms_min": ts[0],
            "top1_full_list_match_ms": full,
            "what": f"TFIDF(top_n=1).match([query], names, re_train=False) against the fitted {len(names)}-name list, "
                    f"host string in, DataFrame out, median of {reps}; top1_full_list_match_ms = "
                    "TFIDF(top_n=1).match(names), the whole self-match"}


def bench_editdistance(args):
    """BASELINE.json config 3 / SURVEY.md §8d: EditDistance (rapidfuzz.fuzz.ratio = Indel ratio) all pairs of
    20 000 x 20 000 IMDB titles (default_rng(0) permutation, first from-title 'Polly Blue Eyes'), first arg-max
    per from-title.  One step = one pass of K4 over all pairs with both lists and the to-side plan resident."""
    import polyfuzz_amd
    from polyfuzz_amd import _lib, datasets
    from polyfuzz_amd.models import EditDistance
    ctx = polyfuzz_amd.Context.default()
    fl, tl = datasets.c3_lists()
    f, t = _lib.DeviceStrings.upload(ctx, fl), _lib.DeviceStrings.upload(ctx, tl)
    plan = _lib.indel_plan_info(ctx, t)
    for _ in range(args.warmup):
        idx, score = _lib.indel_argmax(ctx, f, t)
    ctx.sync()
    ctx.prof_enable(True)
    ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        idx, score = _lib.indel_argmax(ctx, f, t)
    ctx.sync()
    wall = time.perf_counter() - t0
    ctx.prof_enable(False)
    k4_ms, k4_launches = ctx.prof_get("k4_indel")
    k4_step_s = k4_ms / args.steps * 1e-3
    # algorithmic work: one 5-operation word update (u = V & M; V = (V + u) | (V ^ u), + the table look-up) per
    # to-character per 32/64-bit word of the from-string -- counted as 32-bit integer operations
    words32 = np.array([(1 if len(a) <= 32 else 2 * ((len(a) + 63) // 64)) for a in fl], np.float64)
    int_ops = 5.0 * float(words32.sum()) * float(plan["char_steps"])
    cells = float(sum(map(len, fl))) * float(sum(map(len, tl)))
    peak = 256 * 4 * 32 * 2.4e9 / 1e12       # 32-bit integer issue, Tera-op/s (256 CU x 4 SIMD x 32 lanes x 2.4 GHz)
    import oracle
    oracle.build_native()
    rows = 40
    c0 = time.perf_counter()
    e_idx, e_score = oracle.indel_argmax(fl, tl, rows=(0, rows))
    dt = time.perf_counter() - c0
    rows2 = int(max(rows, min(len(fl), rows * args.cpu_seconds / max(dt, 1e-6))))
    c0 = time.perf_counter()
    e_idx, e_score = oracle.indel_argmax(fl, tl, rows=(0, rows2))
    dt = time.perf_counter() - c0
    m = EditDistance(normalize=False)
    m.match(fl, tl)
    ts, ts2 = [], []
    for _ in range(7):
        c0 = time.perf_counter()
        m.match(fl, tl)
        ts.append((time.perf_counter() - c0) * 1e3)
        c0 = time.perf_counter()
        m.match(fl, tl, re_train=False)
        ts2.append((time.perf_counter() - c0) * 1e3)
    out = {
        "metric": "string-pairs/sec, EditDistance (Indel ratio) all pairs + first arg-max, 20k x 20k IMDB titles",
        "value": float(len(fl)) * float(len(tl)) * args.steps / wall, "unit": "pairs/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32/int64 bit-vectors, f64 score",
        "data": "real: reference data/movie_titles.json (IMDB), gzipped in polyfuzz_amd/data/",
        "config": {"workload": "EditDistance(scorer=fuzz.ratio).match(from, to): 20000 x 20000 IMDB titles "
                               "(SURVEY.md §8d config 3), lists and to-side plan resident", "n_from": len(fl), "n_to": len(tl),
                   "alphabet": plan["n_symbols"], "dp_cells": cells, "to_char_steps": plan["char_steps"]},
        "kernel_ms_per_step": {"k4_indel": round(k4_ms / args.steps, 4), "launches_per_step": k4_launches / args.steps},
        "roofline": {"kernel": "k4_indel", "bound": "int32 VALU issue (+ LDS look-ups)", "achieved": int_ops / k4_step_s / 1e12,
                     "peak": peak, "unit": "Tera int-op/s", "frac": int_ops / k4_step_s / 1e12 / peak, "traffic": None,
                     "algorithmic_int_ops_per_step": int_ops, "dp_cell_updates_per_s": cells / k4_step_s,
                     "what": "5 integer operations per to-character per 32-bit word of the from-string (bit-parallel LCS), "
                             "against 256 CU x 4 SIMD x 32 lanes x 2.4 GHz"},
        "cpu_baseline": {"value": rows2 * float(len(tl)) / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
                         "sample": f"first {rows2} from-titles x all {len(tl)} to-titles, oracle/indel.c (plain O(|a||b|) LCS "
                                   f"DP), {dt:.1f} s on 1 host core"},
        "parity_check": {"rows_checked": rows2, "bit_exact": bool(np.array_equal(idx[:rows2], e_idx) and
                                                                  np.array_equal(score[:rows2], e_score))},
        "match_wall_ms": sorted(ts)[len(ts) // 2], "match_wall_ms_to_list_resident": sorted(ts2)[len(ts2) // 2],
        "match_what": "EditDistance(normalize=False).match(from, to): Python lists in, DataFrame out; "
                      "..._to_list_resident = match(from, to, re_train=False), the to-list and its K4 plan kept on the device",
    }
    print(json.dumps(out))


def bench_dense(args):
    """BASELINE.json config 5 / SURVEY.md section 8d: dense cosine top-10 of 500k x 500k 768-d embeddings on 8 GPUs --
    here ONE GPU's share: a 62 500-row from-shard against all 500 000 to-vectors (K5), operands resident in HBM.
    One step = the shard's GEMM panels + row top-n.  Roofline: exact-fp32 MFMA."""
    import polyfuzz_amd
    from polyfuzz_amd import pipeline
    ctx = polyfuzz_amd.Context.default()
    n_to, n_from, d, top_n = 500_000, 62_500, 768, 10
    rng = np.random.default_rng(7)
    b = rng.standard_normal((n_to, d), dtype=np.float32)
    a = rng.standard_normal((n_from, d), dtype=np.float32)
    pick = rng.choice(n_to, n_from, replace=False)
    a += 2.0 * b[pick]                       # planted near-duplicates: the top rank is known
    job = pipeline.DenseMatchJob(ctx, a, b, top_n=top_n)
    for _ in range(max(1, args.warmup)):
        job.step()
    ctx.sync()
    ctx.prof_enable(2)
    ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = job.step()
    ctx.sync()
    wall = time.perf_counter() - t0
    ctx.prof_enable(False)
    gemm_ms, launches = ctx.prof_get("k5_gemm_panel")
    idx, val = res.download()
    flop = 2.0 * n_from * n_to * d
    # CPU arm + parity on a bounded sample: float64 BLAS cosine + top-n of a few from-rows (oracle/dense.py)
    import oracle
    rows = rng.choice(n_from, 8, replace=False)
    c0 = time.perf_counter()
    bad, err = 0, 0.0
    for i in rows:
        e_idx, e_val = oracle.dense_cossim_topn(a[i:i + 1], b, top_n, 0.0)
        err = max(err, float(np.abs(val[i] - e_val[0]).max()))
        bad += int(not np.array_equal(idx[i], e_idx[0]))
    dt = time.perf_counter() - c0
    gemm_s = gemm_ms / max(1, launches) * 1e-3 * (launches / args.steps)       # GEMM time per step
    out = {
        "metric": "vector pairs/sec, dense cosine top-10, one GPU's shard of 500k x 500k x 768 (BASELINE config 5)",
        "value": float(n_from) * n_to * args.steps / wall, "unit": "pairs/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic (SURVEY.md section 8d config 5: standard normal rows, planted near-duplicates)",
        "config": {"workload": "Embeddings-style cosine top-10: 62 500 from-vectors (one of 8 row shards) x 500 000 to-vectors x 768, "
                               "operands resident, pipeline.DenseMatchJob", "n_from": n_from, "n_to": n_to, "dim": d, "top_n": top_n},
        "kernel_ms_per_step": {"k5_gemm_panel": round(gemm_ms / args.steps, 3), "launches_per_step": launches / args.steps},
        "roofline": {"kernel": "k5_gemm_panel_pipe", "bound": "mfma", "achieved": flop / gemm_s / 1e12, "peak": 157.3,
                     "unit": "TFLOP/s", "frac": flop / gemm_s / 1e12 / 157.3, "traffic": None,
                     "end_to_end_frac": flop * args.steps / wall / 1e12 / 157.3,
                     "what": "2 n_from n_to d flops of exact fp32 products (v_mfma_f32_32x32x2_f32) over the GEMM panels' "
                             "summed launch time; end_to_end_frac = over the whole step (row top-n included)"},
        "cpu_baseline": {"value": len(rows) * float(n_to) / dt, "unit": "pairs/s", "cores": int(os.cpu_count() or 1), "kind": "port",
                         "sample": f"{len(rows)} from-rows x all {n_to} to-vectors, oracle/dense.py (float64 numpy / BLAS), {dt:.1f} s"},
        "parity_check": {"rows_checked": int(len(rows)), "rows_with_index_diff": bad, "max_abs_score_err": err,
                         "planted_match_found_top1": float((idx[:, 0] == pick).mean())},
    }
    print(json.dumps(out))


def main():
    args = parse()
    if args.config == "editdistance":
        if args.gpus != 1:
            raise SystemExit("--config editdistance is a single-GPU configuration")
        return bench_editdistance(args)
    if args.config == "dense":
        if args.gpus != 1:
            raise SystemExit("--config dense times one GPU's shard; the 8-GPU job is eight of them (pipeline.DenseMatchJob)")
        return bench_dense(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    dist = torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import polyfuzz_amd
    from polyfuzz_amd import _lib, pipeline, synth

    ctx = polyfuzz_amd.Context(local_rank)
    info = ctx.info()
    comm, exchange = None, "none (single GPU)"
    if world > 1:
        # the library's own RCCL communicator (bootstrap: broadcast of the 128-byte id through torch)
        err = ""
        try:
            comm = _lib.Comm.from_torch_distributed(ctx, dist)
        except Exception as e:       # keep every rank in step: agree on the outcome before going on
            err = f"{type(e).__name__}: {e}"
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            exchange = "RCCL all-gather of the per-shard top-n blocks (the fit runs on the replicated list: no exchange)"
        else:
            if comm is not None:
                comm.free()
            comm = None
            exchange = f"DISABLED -- library RCCL communicator failed ({err or 'on another rank'}); ranks ran as independent replicas"

    # ---- inputs (resident in HBM before timing) ----
    names, kind = the_list(args)
    n = len(names)
    if world == 1:
        job = pipeline.TfidfMatchJob(ctx, names, None, top_n=args.top_n, min_similarity=MIN_SIM, self_match=True)
        n_from_total, shard_desc = n, "the whole list"
    elif args.scaling == "strong":
        b, e = pipeline.shard_bounds(n, world, rank)
        rpr = pipeline.shard_bounds(n, world, 0)[1]
        job = pipeline.TfidfMatchJob(ctx, names[b:e], names, top_n=args.top_n, min_similarity=MIN_SIM, self_match=True,
                                     shard_offset=b, comm=comm, rows_per_rank=rpr)
        n_from_total, shard_desc = n, f"rows [{b}, {e}) of the list"
    else:
        shard = names if rank == 0 else synth.company_names(n, seed=1234 + rank)
        job = pipeline.TfidfMatchJob(ctx, shard, names, top_n=args.top_n, min_similarity=MIN_SIM, self_match=True,
                                     shard_offset=rank * n, comm=comm, rows_per_rank=n)
        n_from_total = n * world
        shard_desc = "rank 0: the list itself, rank r > 0: synthetic names of the same token statistics"

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
            ctx.sync()

    for _ in range(args.warmup):
        job.step()
    barrier()
    ctx.prof_enable(2)          # live HIP-event timing of every K3 launch (the dominant kernel) inside the timed region
    ctx.prof_reset()
    t0 = time.perf_counter()
    ctx.event_record(0)
    for _ in range(args.steps):
        result = job.step()
    ctx.event_record(1)
    barrier()
    wall = time.perf_counter() - t0
    ctx.prof_enable(False)
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # the per-kernel breakdown of a step: three more steps with every profiled kernel bracketed, outside the timed region
    k3_timed = ctx.prof_get("k3_cossim_topn") if rank == 0 else None
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(3):
        result = job.step()
    barrier()
    ctx.prof_enable(False)
    out = None
    if rank == 0:
        kernel_ms = {name: round(ctx.prof_get(name)[0] / 3, 4) for name in pipeline.PROFILED_KERNELS}
        stats = job.stats()
        k3_ms, k3_launches = k3_timed
        gpu_ms = ctx.event_elapsed_ms(0, 1)
        pairs_per_step = float(n_from_total) * float(n)
        value = pairs_per_step * args.steps / wall
        k3_avg_s = (k3_ms / max(1, k3_launches)) * 1e-3
        ix = job.index.info()
        # algorithmic bytes of one K3 launch (SURVEY.md §8d / DESIGN.md): one 8-byte posting per
        # multiply-add + the from-side CSR once + the (idx, score) results once
        bytes_alg = 8.0 * stats["madds"] + 8.0 * stats["nnz_from"] + 8.0 * job.n_from * args.top_n
        achieved = bytes_alg / k3_avg_s / 1e9 if k3_avg_s > 0 else 0.0
        # what the kernel is really bound by: LDS.  Floor = every multiply-add is one ds_add_u32 lane (measured
        # atomic rate) + every accumulator cell of every (from-row, to-block) is read and cleared once
        cells = float(job.n_from) * ix["n_blocks"] * ix["block_cols"]
        lds_floor_s = stats["madds"] / LDS_ATOMIC_LANES_PER_S + cells * 8.0 / (N_CU * LDS_BYTES_PER_CLK_CU * CLK_HZ)
        traffic, traffic_note = recorded_traffic(args)
        out = {
            "metric": "string-pairs/sec, TF-IDF cosine top-n 100k x 100k (value = device-resident step; "
                      "match_wall_ms = .match() list -> DataFrame; latency.top1_single_query_ms = top-1 match latency)",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "real: reference data/company_names.json (100 000 SEC-EDGAR names, gzipped in polyfuzz_amd/data/)"
                    if kind == "real" else "synthetic",
            "config": {
                "workload": f"TFIDF(min_similarity={MIN_SIM}, top_n={args.top_n}).match(names): self-match of "
                            f"{'the first ' + str(n) + ' of the ' if n < N_NAMES else 'all '}"
                            f"{N_NAMES if kind == 'real' else n} {kind} company names, char-3-gram TF-IDF cosine "
                            f"(SURVEY.md §8d headline; reference docs/tutorial/datasets/datasets.md:36-41)",
                "n_from_total": n_from_total, "n_from_this_rank": job.n_from, "n_to": n, "top_n": args.top_n,
                "from_rows": shard_desc,
                "vocab": stats["vocab"], "nnz_from": stats["nnz_from"], "nnz_to": stats["nnz_to"],
                "multiply_adds_rank0": stats["madds"],
                "step": job.step_description(),
                "parallelism": f"from-rows sharded x{world}, list replicated",
                "exchange": exchange,
                "device": info["name"],
            },
            "gpu_ms_per_step_rank0": gpu_ms / args.steps,
            "kernel_ms_per_step": kernel_ms,
            "roofline": {
                "kernel": "k3_cossim_topn",
                "bound": "lds",
                "bound_note": "the contract's figure (achieved/peak/frac) prices the ALGORITHMIC bytes against the HBM "
                              "peak; the postings are served by L2 / Infinity Cache, so HBM is not the limiter -- the "
                              "kernel is bound by LDS atomics + the accumulator sweep and by instruction issue: see "
                              "lds_floor_ms / frac_of_lds_floor",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_note": traffic_note or "no PMC record for this workload (profiles/k3_hbm_traffic.json)",
                "algorithmic_bytes_per_launch": bytes_alg,
                "avg_launch_ms": k3_avg_s * 1e3,
                "launches": k3_launches,
                "lds_floor_ms": lds_floor_s * 1e3,
                "frac_of_lds_floor": lds_floor_s / k3_avg_s if k3_avg_s > 0 else 0.0,
                "lds_floor_what": f"{stats['madds']:.4g} ds_add_u32 lanes at {LDS_ATOMIC_LANES_PER_S:.1e}/s + "
                                  f"{cells:.4g} accumulator cells x 8 B (read + clear) at "
                                  f"{N_CU * LDS_BYTES_PER_CLK_CU * CLK_HZ / 1e12:.1f} TB/s of LDS bandwidth",
            },
        }
        idx = val = None
        if world == 1:
            idx, val = result.download()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["cpu_baseline_arms"], out["parity_check"] = \
                cpu_baselines_and_check(job, idx, val, args.cpu_seconds, names)
        if world == 1 and not args.no_match_wall:
            # (profiling is off again: these launches do not enter the K3 average above)
            mw, m_fit = match_wall(names, args.top_n, idx)
            out.update(mw)
            out["latency"] = top1_latency(m_fit, names)
    barrier()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
