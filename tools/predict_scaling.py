"""PREDICTION of the 1 -> 8 GPU curve on ONE GPU (no multi-GPU node was available to any round): every rank's shard of an
N-GPU job is run here, one after the other, exactly as `bench.py --gpus N` would build it (same shard bounds, same lists,
to-side replicated), and timed; a job's step time is the slowest rank's (bench.py takes the max over ranks) plus the
all-gather of the results, which is PRICED, not measured (ring over xGMI: (N-1)/N of the gathered bytes through one
153 GB/s link).  Shards are independent (no data-path collective before the gather), so what bounds
the curve is the balance of the shards: max / mean of the per-rank times.

Round 5: the headline's strong-scaling job runs K3's symmetric form cut over the ranks (rank r: the rows r, r + N, ...;
pfz_comm_cossim_topn_symmetric).  Its ranks are NOT independent (two all-gathers), so a rank cannot be run alone through the
product path; the variant library (tools/build_variant.sh variants/exp.so -DPFZ_EXPERIMENTS) has a knob that runs ONE part's pass 1
and magnet items alone (PFZ_K3_SYM_SOLO=p/N; pass 0 of all rows, the re-deal, the merge and pass 2 as on one GPU) -- see
`symmetric_strong` below for what is measured and what is priced.

usage: POLYFUZZ_HIP_LIB=$PWD/variants/exp.so python tools/predict_scaling.py [--quick] > profiles/r05_predicted_scaling.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import polyfuzz_amd
from polyfuzz_amd import datasets, pipeline, synth

XGMI_LINK_GBS = 153.0
QUICK = "--quick" in sys.argv
ctx = polyfuzz_amd.Context.default()


def timed(step, reps=3, warm=1):
    for _ in range(warm):
        step()
    ctx.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        step()
        ctx.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


def gather_ms(n_rows_total, top_n, world):
    if world == 1:
        return 0.0
    return (world - 1) / world * n_rows_total * top_n * 8 / (XGMI_LINK_GBS * 1e9) * 1e3


def summarise(name, t1, per_world, top_n, rows_total_of, scaling):
    rec = {"config": name, "scaling": scaling, "single_gpu_ms": t1, "worlds": {}}
    for world, ts in per_world.items():
        g = gather_ms(rows_total_of(world), top_n, world)
        job = max(ts) + g
        eff = (t1 / job) if scaling == "weak" else (t1 / (world * job))
        rec["worlds"][str(world)] = {"per_rank_ms": [round(t, 3) for t in ts], "max_over_mean": round(max(ts) / (sum(ts) / len(ts)), 4),
                                     "allgather_ms_priced": round(g, 4), "job_ms_predicted": round(job, 3),
                                     "efficiency_predicted": round(eff, 4)}
    return rec


def tfidf_job(from_shard, to_list, top_n, self_match, offset, cuts=None):
    job = pipeline.TfidfMatchJob(ctx, from_shard, to_list, top_n=top_n, min_similarity=0.0, self_match=self_match, shard_offset=offset)
    return timed(job.step)


out = {"what": __doc__.split("\n\n")[0], "device": ctx.info()["name"], "measured_on": "ONE MI355X, shards run sequentially",
       "configs": []}
names = datasets.load_company_names()
n = len(names)
worlds = (2, 8) if QUICK else (2, 4, 8)

# ---- headline, weak scaling (the driver's `bench.py --gpus N`): rank 0 the list itself, rank r synthetic names ---------
t1 = tfidf_job(names, None, 5, True, 0)
per = {}
shard_ms = {0: tfidf_job(names, names, 5, True, 0)}
for r in range(1, max(worlds)):
    shard_ms[r] = tfidf_job(synth.company_names(n, seed=1234 + r), names, 5, True, r * n)
for w in worlds:
    per[w] = [shard_ms[r] for r in range(w)]
out["configs"].append(summarise("headline 100k x 100k top-5, weak (every rank 100k from-rows, list replicated)", t1, per, 5,
                                lambda w: n * w, "weak"))

# ---- headline, strong scaling: contiguous row shards of the (sorted, skewed) list; equal-rows vs cost-balanced cuts ------
for label, bounds_of in (("equal row counts (pipeline.shard_bounds)", lambda w: [pipeline.shard_bounds(n, w, r) for r in range(w)]),
                         ("cost-balanced cuts (pipeline.balanced_bounds)", lambda w: pipeline.balanced_bounds(names, w))):
    per = {}
    for w in worlds:
        per[w] = [tfidf_job(names[b:e], names, 5, True, b) for b, e in bounds_of(w)]
    rec = summarise(f"headline 100k x 100k top-5, strong, {label}", t1, per, 5, lambda w: n, "strong")
    rec["bounds_8"] = [list(map(int, be)) for be in bounds_of(8)]
    out["configs"].append(rec)

# ---- headline, strong scaling, K3's symmetric form cut over the ranks (what bench.py --gpus N runs by default since round 5) ----
PASS0_MS = 0.199          # k3_sym_kernel<2048, 0> of all 100 000 rows (profiles/experiments/r05_k3_slot_order_steps.txt, run f)
if os.environ.get("POLYFUZZ_HIP_LIB", "").endswith("exp.so") and n >= 20480:
    def ring_ms(bytes_per_rank, world):
        return (world - 1) * bytes_per_rank / (XGMI_LINK_GBS * 1e9) * 1e3
    job1 = pipeline.TfidfMatchJob(ctx, names, None, top_n=5, min_similarity=0.0, self_match=True)
    os.environ.pop("PFZ_K3_SYM_SOLO", None)
    t_sym1 = timed(job1.step)
    rec = {"config": "headline 100k x 100k top-5, strong, K3's symmetric form cut over the ranks (rows r, r + N, ...)", "scaling": "strong",
           "single_gpu_ms": t_sym1, "worlds": {},
           "how": "per rank: the WHOLE step measured on one GPU with only that part's pass 1 + magnet items (variant library, "
                  "PFZ_K3_SYM_SOLO=p/N) minus the pass-0 time of the other parts' rows (pass 0 is dealt over the ranks in the real "
                  f"job: {PASS0_MS} ms x (1 - 1/N)); plus, PRICED as rings over one 153 GB/s xGMI link: the in-place all-gather of the "
                  "thresholds (n x 4 B in all) and the all-gather of the candidate lists (n x top_n x 8 B per rank); plus the merge of "
                  "the N lists per row (measured once: the world-1 product path's k3_sym_merge_parts is part of single_gpu_ms's sibling "
                  "below).  fit + vectorise + index are replicated (every rank does the whole list's)."}
    for w in worlds:
        ts = []
        for p in (range(w) if w <= 4 else (0, w // 2, w - 1)):
            os.environ["PFZ_K3_SYM_SOLO"] = f"{p}/{w}"
            ts.append(timed(job1.step) - PASS0_MS * (1.0 - 1.0 / w))
        os.environ.pop("PFZ_K3_SYM_SOLO", None)
        comm = ring_ms(n * 4 / w, w) + ring_ms(n * 5 * 8, w)
        job = max(ts) + comm
        rec["worlds"][str(w)] = {"per_rank_ms": [round(t, 3) for t in ts], "max_over_mean": round(max(ts) / (sum(ts) / len(ts)), 4),
                                 "allgathers_ms_priced": round(comm, 4), "job_ms_predicted": round(job, 3),
                                 "efficiency_predicted": round(t_sym1 / (w * job), 4)}
    out["configs"].append(rec)

# ---- config 4: 1M x 1M top-10 on 8 GPUs = 8 shards of 125 000 from-rows against the replicated 1M to-list ---------------
if not QUICK:
    to_1m = synth.company_names(1_000_000, 5678)
    ts = []
    for r in range(8):
        ts.append(tfidf_job(synth.company_names(125_000, 1234 + r), to_1m, 10, False, 0))
    out["configs"].append(summarise("config 4: 1M x 1M top-10 over 8 GPUs (125 000 from-rows per rank, to-list replicated)",
                                    sum(ts), {8: ts}, 10, lambda w: 1_000_000, "strong"))
    del to_1m

# ---- config 3 / RapidFuzz: 20k x 20k titles, strong --------------------------------------------------------------------
fl, tl = datasets.c3_lists()
for scorer in ("ratio", "WRatio"):
    job1 = pipeline.BestChoiceJob(ctx, fl, tl, scorer=scorer)
    t1 = timed(job1.step)
    per = {}
    for w in worlds:
        ts = []
        for r in range(w):
            b, e = pipeline.shard_bounds(len(fl), w, r)
            ts.append(timed(pipeline.BestChoiceJob(ctx, fl[b:e], tl, scorer=scorer).step))
        per[w] = ts
    out["configs"].append(summarise(f"config 3: 20k x 20k IMDB titles, {scorer}, strong", t1, per, 2, lambda w: len(fl), "strong"))

# ---- config 5: dense 500k x 500k x 768 on 8 GPUs = 62 500 from-vectors per rank; two of the eight shards ----------------
if not QUICK:
    rng = np.random.default_rng(1)
    to_v = rng.standard_normal((500_000, 768), dtype=np.float32)
    ts = []
    for r in range(2):
        fv = np.random.default_rng(100 + r).standard_normal((62_500, 768), dtype=np.float32)
        job = pipeline.DenseMatchJob(ctx, fv, to_v, top_n=10)
        ts.append(timed(job.step, reps=2))
        del job
    out["configs"].append({"config": "config 5: dense 500k x 500k x 768 top-10 over 8 GPUs (62 500 from-vectors per rank)",
                           "scaling": "strong", "per_rank_ms_of_2_of_8_shards": [round(t, 2) for t in ts],
                           "note": "dense shards do the same flops whatever their rows: max / mean = 1 by construction; the "
                                   "two shards measured differ by clock noise only",
                           "allgather_ms_priced": round(gather_ms(500_000, 10, 8), 4),
                           "job_ms_predicted": round(max(ts) + gather_ms(500_000, 10, 8), 2),
                           "efficiency_predicted": round(sum(ts) / len(ts) / (max(ts) + gather_ms(500_000, 10, 8)), 4)})
print(json.dumps(out, indent=1))
