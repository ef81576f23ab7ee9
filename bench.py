#!/usr/bin/env python3
"""bench.py -- string-pairs/sec + top-1 match latency of the TF-IDF cosine top-n hot path on MI355X, with every other
BASELINE.json configuration measured beside it in the same run.

Headline workload (BASELINE.json metric "TF-IDF cosine 100k x 100k"; SURVEY.md section 8d "Headline"): the 100 000 real
SEC-EDGAR company names of the reference's data/company_names.json (shipped gzipped in polyfuzz_amd/data/, the GPU
box has no network), SELF-MATCH, char-3-gram TF-IDF, cosine top-5, min_similarity 0 -- i.e.
`TFIDF(min_similarity=0, top_n=5).match(names)` (reference docs/tutorial/datasets/datasets.md:36-41).

One "step" (the default, --step match) = ONE USER-LEVEL CALL: `TFIDF(min_similarity=0, top_n=5).match(names)`, Python list in,
DataFrame out -- host packing, H2D, fit vocabulary + idf on the list, vectorise it, build the inverted index, the fused cosine
top-n with the diagonal excluded, the results to the host, the frame (reference _tfidf.py:68-118, _utils.py:54-125).  That is the
unit SURVEY.md section 8d defines the metric on ("pairs/s = N_from * N_to / wall time of .match()"), and round 5's review asked
for it as THE value: `value` = N_from * N_to * steps / wall of those calls.  The frames of the timed calls are kept until the
clock has stopped, on a heap the process has touched before (FrameKeeper: the disposal of a result -- ~0.9 ms of reference counting
-- and the first touch of fresh pages are not part of a call); `match_wall_ms` is the same call under a per-call clock (median of 7).
The bench contract's own reading of `value` -- the same work with the list ALREADY RESIDENT IN HBM, nothing crossing PCIe inside
the timed region (fit + vectorise + index + K3: `TfidfMatchJob.step`) -- is timed in a second region of the same run, same
protocol, and reported as `device_step` (`--step device` makes it the line); `kernel_ms_per_step` and `roofline` (the dominant
kernel, K3, timed live with HIP events on the library's stream) are that region's.

The same JSON line also carries, measured after the timed regions on rank 0 at N = 1:
  match_wall_ms / match_stages_ms    the same call under a median-of-7 protocol with its stage stamps
  latency.top1_single_query_ms       one query string against the fitted 100k list, top-1, re_train=False
  roofline, cpu_baseline (+ cpu_baseline_arms), parity_check (EVERY from-row the all-cores CPU arm computes -- the whole list on a
                                     many-core box -- vs the oracle, from the strings up)
  configs                            compact sub-records, each with ms_per_step, match_wall_ms, roofline, cpu_baseline and
                                     a parity_check over every row its CPU arms compute:  c2_tfidf_10k (config 2), editdistance
                                     (config 3), rapidfuzz_wratio (RapidFuzz() = what PolyFuzz("EditDistance") runs, on config 3's
                                     lists), dense_shard (one GPU's share of config 5), tfidf_1m_shard (of config 4)

Multi-GPU (--gpus N: `python bench.py --gpus N` launches torch.distributed.run itself, one process per GPU, rendezvous over
gloo, RCCL the library's alone -- or `--transport local`: one process, N contexts, one host thread per rank, the library's
in-process transport: a rehearsal of the same rank logic on however many GPUs are visible, down to one):
  --config tfidf         --scaling strong (default at N > 1): the one 100k x 100k self-match cut over the ranks (K3's symmetric
                         form: every unordered pair once over all ranks, the ranks' candidate lists all-gathered and merged);
                         the timed step is the same user-level call on every rank (pipeline.sharded_self_match);
                         --scaling weak: every rank matches its own batch of 100k from-rows (the real names in a rank-seeded
                         random order) against the replicated real list, a two-list match
  --config dense         weak: 62 500 from-vectors per rank against 500 000 replicated to-vectors (N = 8 IS config 5)
  --config editdistance  strong: config 3's 20 000 from-titles split over the ranks
  --config rapidfuzz     strong: the same lists under RapidFuzz's default scorer (WRatio)
The data-path exchanges are all-gathers: of the per-shard result blocks, or -- the self-match in K3's symmetric form -- of the
ranks' pass-0 thresholds and per-row candidate lists.  torch is used for rendezvous / barrier /
the max-over-ranks only, never in the data path.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
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
LDS_ATOMIC_LANES_PER_S = 4.059e12  # measured ds_add_u32 rate, all CUs: profiles/r04_ubench/lds_atomic.txt (tools/ubench/lds_atomic.hip, 6.61 lanes/clk/CU)
INT32_PEAK_TOPS = N_CU * 4 * 32 * CLK_HZ / 1e12      # 32-bit integer issue: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T/s
FP32_MFMA_PEAK_TFLOPS = 157.3
SEED = 20260924                  # of every random row sample below


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=N_NAMES,
                    help="list length: <= 100000 takes the first n real names, more takes synthetic names")
    ap.add_argument("--top-n", type=int, default=TOP_N)
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="default: weak for tfidf / dense, strong for editdistance / rapidfuzz")
    ap.add_argument("--transport", choices=("rccl", "local"), default="rccl",
                    help="local: one process, --gpus contexts (round-robin over the visible devices), host threads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-match-wall", action="store_true", help="skip the .match() wall time and the latency leg")
    ap.add_argument("--no-configs", action="store_true", help="headline only: skip the `configs` sub-records")
    ap.add_argument("--small", action="store_true",
                    help="rehearsal sizes (tests of the launch / rank logic): 2 000 x 2 000 titles, 4 000 x 20 000 x 256 vectors; "
                         "the line says so in config.rehearsal")
    ap.add_argument("--step", choices=("match", "device"), default="match",
                    help="what the headline's timed step is.  match (default): the user-level call of SURVEY section 8d's metric -- "
                         "TFIDF(min_similarity=0, top_n=5).match(names), Python list in, DataFrame out (N > 1: pipeline.sharded_self_match "
                         "on every rank) -- with the device-resident step timed in a second region (`device_step`); device: the "
                         "device-resident step alone (list resident in HBM: fit + vectorise + index + K3)")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="tests only (tests/test_bench_launch_cpu.py): the launch / rendezvous / rank logic of --gpus N on a box "
                         "WITHOUT a GPU -- hands the ranks to tests/bench_rehearsal.py (TfidfMatchJob through tests/cpu_engine.py on a "
                         "400-name list); prints a rehearsal record, never a bench line")
    ap.add_argument("--config", choices=("tfidf", "c2", "editdistance", "rapidfuzz", "dense", "tfidf_1m"), default="tfidf",
                    help="tfidf: the headline (+ every other config as a sub-record at N = 1); the others: that "
                         "configuration alone as the line")
    return ap.parse_args(argv)


# ---- who am I: single process, one process per GPU (torch.distributed), or one thread per rank ------------------

class World:
    """rank / size + the two collectives the bench itself needs (barrier, max of a host float)."""
    rank, size, kind = 0, 1, "single"

    def barrier(self, ctx):
        ctx.sync()

    def max(self, x):
        return x

    def comm(self, ctx):
        """(communicator, what it exchanges) -- made once per world"""
        if getattr(self, "_comm", None) is None:
            self._comm = self._make_comm(ctx)
        return self._comm

    def _make_comm(self, ctx):
        return None, "none (single GPU)"


class TorchWorld(World):
    """One process per GPU.  torch.distributed is the RENDEZVOUS only, over gloo (host TCP): the id of the library's communicator is
    broadcast through it, the bench's own barrier and max-over-ranks run on it.  Round 6: not over torch's
    "nccl" backend -- that made torch a second RCCL user in the process beside the library's communicator -- and the library is
    loaded BEFORE torch, so that the librccl it was linked against (/opt/rocm) is the one it runs on, not the copy inside the torch
    wheel; `config.rccl` in the line says which one the process resolved."""
    kind = "rccl"

    def __init__(self, need_devices=True):
        from polyfuzz_amd import _lib
        _lib.load()                       # (first: the library's own librccl.so.1, see above)
        self.rccl = _lib.rccl_versions()
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        dist.init_process_group("gloo")
        # every rank needs a device of its own: agree on that before anybody creates a context
        n_dev = _lib.device_count()
        t = torch.tensor([n_dev], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        print(f"[bench rank {self.rank}] rendezvous ok (gloo, world {self.size}); {n_dev} device(s) visible; librccl "
              f"{self.rccl['runtime']} at {self.rccl['path']} (library compiled against rccl.h {self.rccl['header']})",
              file=sys.stderr, flush=True)
        if need_devices and int(t.item()) < self.size:
            dist.barrier()
            dist.destroy_process_group()
            raise SystemExit(f"[bench rank {self.rank}] --gpus {self.size}: {self.size} devices needed, {int(t.item())} visible")
        if self.rccl["header"] // 10000 != self.rccl["runtime"] // 10000:
            raise SystemExit(f"[bench rank {self.rank}] librccl {self.rccl['runtime']} at {self.rccl['path']} does not match the rccl.h "
                             f"{self.rccl['header']} the library was compiled against (major versions differ)")

    def barrier(self, ctx):
        ctx.sync()
        self.dist.barrier()
        ctx.sync()

    def max(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def _make_comm(self, ctx):
        """the library's own RCCL communicator (bootstrap: broadcast of the 128-byte id through the gloo group)"""
        from polyfuzz_amd import _lib
        err, comm = "", None
        try:
            comm = _lib.Comm.from_torch_distributed(ctx, self.dist)
        except Exception as e:       # keep every rank in step: agree on the outcome before going on
            err = f"{type(e).__name__}: {e}"
        ok = self.torch.tensor([0 if err else 1], dtype=self.torch.int32)
        self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            return comm, "RCCL all-gather of the per-shard result blocks"
        if comm is not None:
            comm.free()
        return None, f"DISABLED -- library RCCL communicator failed ({err or 'on another rank'}); ranks ran as independent replicas"

    def close(self):
        self.dist.destroy_process_group()


def self_launch(gpus):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment (how the driver calls the bench): become the
    launcher -- the contract's own command line, one rank per GPU --, pass rank 0's JSON line through, return the job's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {gpus} without WORLD_SIZE: launching {' '.join(cmd[1:8])} ... ({gpus} ranks)", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


class LocalWorld(World):
    """N ranks as N host threads of this process, each with its own context (its own stream, allocator, staging
    buffer) on device rank % visible-devices and a communicator of the library's in-process transport."""
    kind = "local"

    def __init__(self, rank, size, shared):
        self.rank, self.size, self.shared = rank, size, shared

    def barrier(self, ctx):
        ctx.sync()
        self.shared["barrier"].wait()

    def max(self, x):
        self.shared["vals"][self.rank] = x
        self.shared["barrier"].wait()
        m = max(self.shared["vals"])
        self.shared["barrier"].wait()
        return m

    def _make_comm(self, ctx):
        return self.shared["comms"][self.rank], ("in-process transport (host rendezvous + device-to-device copies) all-gather "
                                                 "of the per-shard result blocks")


def the_list(args):
    from polyfuzz_amd import datasets, synth
    if args.n <= N_NAMES:
        return datasets.load_company_names()[:args.n], "real"
    return synth.company_names(args.n, seed=5678), "synthetic"


def traffic_key(args, label_key=None):
    """key of a workload in profiles/k3_hbm_traffic.json"""
    if label_key is not None:
        return label_key
    return "headline" if (args.n == N_NAMES and args.top_n == TOP_N) else f"company_names[:{args.n}] self-match top-{args.top_n}"


def recorded_traffic(key):
    """Bytes per K3 launch that missed L2 (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes, tools/pmc_traffic.sh ->
    profiles/k3_hbm_traffic.json; a bench run cannot collect PMC itself); only valid for the workload it was recorded on."""
    try:
        with open(os.path.join(REPO, "profiles", "k3_hbm_traffic.json")) as f:
            rec = json.load(f)["records"][key]
        return float(rec["hbm_bytes_per_launch"]), f"{rec['note']} [{rec['kernel']}, {rec['source']}]"
    except (OSError, KeyError, ValueError, TypeError):
        return None, None


def n_cores():
    """host threads the all-cores CPU arms use: the affinity mask, capped by the cgroup's CPU quota where there is one -- the GPU
    box shows 256 CPUs in its mask and grants 16 CPUs' worth of time (cpu.max 1600000 / 100000, tools/r6_topo_probe.sh): 256 threads
    there are 16 cores thrashing, and "cores: 256" in a CPU arm's record said more than the box gave"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def median(xs):
    return sorted(xs)[len(xs) // 2]


# ---- K3: roofline object, CPU arms, random-row parity ------------------------------------------------------------

def k3_roofline(job, stats, k3_ms, k3_launches, top_n, traffic=None, traffic_note=None):
    """The bench contract's roofline object for K3, the same keys every round:
      bound / achieved / peak / frac   the resource that binds the kernel: every multiply-add the kernel EXECUTES is one
                               ds_add_u32 lane at the measured rate (profiles/r04_ubench/lds_atomic.txt) + every accumulator cell of
                               every (from-row, to-block) read and cleared once at the LDS bandwidth; floor / launch time
                               (== frac_lds_floor), expressed as LDS bandwidth.
      frac_hbm_priced          SURVEY section 8d's ALGORITHMIC bytes of one launch (one 8-byte posting per multiply-add of every
                               ordered pair + the from-side CSR once + the results once) / the kernel's average launch time,
                               against the 8 TB/s HBM peak.  A price: it can exceed 1 (the postings come from L2 / Infinity Cache).
      executed_bytes           the same pricing of the multiply-adds the kernel really does (symmetric form: every unordered pair once)
      traffic                  bytes per launch that left L2 (PMC record, profiles/k3_hbm_traffic.json), or null
      compulsory_bytes         inputs once + results once"""
    k3_avg_s = (k3_ms / max(1, k3_launches)) * 1e-3
    ix = job.index.info()
    n_rows_job = job.n_to if getattr(job, "result_is_full", False) else job.n_from
    bytes_alg = 8.0 * stats["madds"] + 8.0 * stats["nnz_from"] + 8.0 * n_rows_job * top_n
    hbm_priced = bytes_alg / k3_avg_s / 1e9 if k3_avg_s > 0 else 0.0
    cells = float(job.n_from) * ix["n_blocks"] * ix["block_cols"]
    madds_done = stats["madds"]
    sym_launches, sym_rows = job.index.symmetric_launches()
    symmetric = sym_launches > 0 and sym_rows >= n_rows_job and "madds_symmetric" in stats
    if symmetric:
        # a list against itself ran in K3's symmetric form (k3_symmetric.hip): every unordered pair of rows scored once.  The
        # SURVEY figure (`frac_hbm_priced`) prices the job as the reference's library does it -- every ordered pair --; the LDS
        # floor is priced on what this kernel executes
        madds_done, cells = stats["madds_symmetric"], stats["cells_symmetric"]
    if getattr(job, "result_is_full", False):
        # ... cut over the ranks (rows r, r + world, ...): this rank's share of the job
        w = float(job.comm.world)
        madds_done, cells, bytes_alg, hbm_priced = madds_done / w, cells / w, bytes_alg / w, hbm_priced / w
    lds_bw = N_CU * LDS_BYTES_PER_CLK_CU * CLK_HZ
    lds_floor_s = madds_done / LDS_ATOMIC_LANES_PER_S + cells * 8.0 / lds_bw
    # compulsory: the from-side CSR once, the to-side index as it lies in HBM (padded pieces + table) once, the results once
    compulsory = 8.0 * stats["nnz_from"] + 4.0 * (job.n_from + 1) + 8.0 * float(ix["n_pieces"]) * float(ix["piece_postings"]) + float(ix["table_bytes"]) \
        + 8.0 * job.n_from * top_n
    executed_bytes = 8.0 * madds_done + 8.0 * stats["nnz_from"] + 8.0 * job.n_from * top_n
    frac_lds = lds_floor_s / k3_avg_s if k3_avg_s > 0 else 0.0
    return {
        "kernel": "k3_cossim_topn" + (" (symmetric form: k3_sym_kernel passes 0-2 + k3_sym_order / repost / merge)" if symmetric else ""),
        # top level = the resource that binds the kernel: the LDS floor of the work the kernel
        # EXECUTES, as a bandwidth -- frac is a utilisation in (0, 1]
        "bound": "lds", "achieved": frac_lds * lds_bw / 1e9, "peak": lds_bw / 1e9, "unit": "GB/s", "frac": frac_lds,
        "frac_lds_floor": frac_lds,
        # beside it, the contract's HBM pricing of the job (SURVEY section 8d's algorithmic bytes: every ORDERED pair): a price, > 1
        "frac_hbm_priced": hbm_priced / HBM_PEAK_GBS, "hbm_priced_gbs": hbm_priced, "hbm_peak_gbs": HBM_PEAK_GBS,
        "algorithmic_bytes_per_launch": bytes_alg,
        "executed_bytes": executed_bytes,
        "frac_hbm_executed": executed_bytes / k3_avg_s / 1e9 / HBM_PEAK_GBS if k3_avg_s > 0 else 0.0,
        "compulsory_bytes": compulsory,
        "traffic": traffic,
        "traffic_over_algorithmic": (traffic / bytes_alg) if traffic else None,
        "traffic_note": traffic_note or "no PMC record for this workload (profiles/k3_hbm_traffic.json)",
        "lds_floor_ms": lds_floor_s * 1e3, "avg_launch_ms": k3_avg_s * 1e3, "launches": k3_launches,
        "multiply_adds_executed": madds_done, "symmetric_form": bool(symmetric),
        "lds_floor_what": f"{madds_done:.4g} ds_add_u32 lanes at {LDS_ATOMIC_LANES_PER_S:.3e}/s (profiles/r04_ubench/"
                          f"lds_atomic.txt) + {cells:.4g} accumulator cells x 8 B (read + clear) at {lds_bw / 1e12:.2f} TB/s of "
                          f"LDS bandwidth ({N_CU} CU x {LDS_BYTES_PER_CLK_CU:.0f} B/clk x {CLK_HZ / 1e9:.1f} GHz)",
        "bound_note": "bound / achieved / peak / frac: the LDS floor of the executed work (one ds_add_u32 lane per multiply-add at the "
                      "measured rate + every accumulator cell read and cleared once) over the launch time, as LDS bandwidth.  "
                      "frac_hbm_priced = SURVEY 8d's algorithmic bytes (8 B per multiply-add of every ORDERED pair + CSR + results) / "
                      "launch time / 8 TB/s: a PRICE, not a utilisation -- the padded index (tens of MB) is served by L2 / Infinity "
                      "Cache, `traffic` is what left L2; executed_bytes prices the multiply-adds the kernel really does (the "
                      "symmetric form scores every unordered pair once)",
    }


def k3_parity(job, idx, val, rows, e_idx, e_val, a3, b3, n_col, rows_what="seeded random sample of the from-rows"):
    """GPU rows vs the float64 oracle on the same rows: scores within 1e-5; an index mismatch is a real error
    unless the oracle itself has the two scores within 2e-6 (a near-tie fp32 cannot order)."""
    import oracle
    g_idx, g_val = idx[rows], val[rows].astype(np.float64)
    max_err = float(np.abs(g_val - e_val).max()) if len(rows) else 0.0
    mism = np.nonzero((g_idx != e_idx).any(axis=1))[0]
    ties = int((np.diff(e_val, axis=1) == 0).any(axis=1).sum()) if job.top_n > 1 else 0
    hard = 0
    looked = mism[:20000]           # (every differing row has its dense oracle row computed: a result with more of them is wrong anyway)
    for t in looked:
        i = int(rows[t])
        dense = oracle.cossim_dense(a3, b3, n_col, rows=(i, i + 1))[0]
        for r in range(job.top_n):
            if g_idx[t, r] != e_idx[t, r]:
                s = dense[g_idx[t, r]] if g_idx[t, r] >= 0 else 0.0
                if abs(s - e_val[t, r]) >= 2e-6:
                    hard += 1
    return {"rows_checked": int(len(rows)), "rows": rows_what, "max_abs_score_err": max_err,
            "rows_with_index_diff": int(len(mism)), "index_diffs_not_near_ties": hard, "rows_with_exact_ties_in_top_n": ties,
            "ok": bool(max_err <= 1e-5 and hard == 0 and len(looked) == len(mism))}


ORACLE_VECTORISE_PY_MAX = 250_000     # the Python restatement of the vectoriser does ~20 us per string and pass; longer
#                                       lists go through its numpy twin (oracle/tfidf_numpy.py, pinned on it bit for bit)


def oracle_matrices(job, from_strings, to_strings):
    """The lists vectorised by the ORACLE (oracle/tfidf_oracle.py == scikit-learn bit for bit, or its numpy twin for lists
    of more than ORACLE_VECTORISE_PY_MAX strings; reference _tfidf.py:102-118: fit on to + from, or on the one list of a
    self-match) and the device's K1 / K2 output held against it: CSR structure equal, values within one fp32 rounding.
    Returns (a3, b3, n_col, record)."""
    import oracle
    n = len(from_strings) + (0 if to_strings is None else len(to_strings))
    t0 = time.perf_counter()
    if n > ORACLE_VECTORISE_PY_MAX:
        which = "oracle/tfidf_numpy.py (== oracle/tfidf_oracle.py == sklearn TfidfVectorizer bit for bit)"
        o = oracle.TfidfNumpyOracle()
        if to_strings is None:
            o.fit(from_strings)
            a3 = b3 = o.transform_fitted(0, n)
        else:
            o.fit(list(to_strings) + list(from_strings))
            b3, a3 = o.transform_fitted(0, len(to_strings)), o.transform_fitted(len(to_strings), n)
        n_vocab = len(o.codes)
    else:
        which = "oracle/tfidf_oracle.py (== sklearn TfidfVectorizer bit for bit)"
        o = oracle.TfidfOracle()
        if to_strings is None:
            o.fit(from_strings)
            a3 = b3 = o.transform(from_strings)
        else:
            o.fit(list(to_strings) + list(from_strings))
            a3, b3 = o.transform(from_strings), o.transform(to_strings)
        n_vocab = len(o.vocabulary)
    dt = time.perf_counter() - t0
    d_a, d_b, n_col = job.host_matrices()
    rec = {"what": f"device CSR (K1/K2) vs {which} on the same lists",
           "strings": n, "vocab_device": int(n_col), "vocab_oracle": n_vocab, "oracle_seconds": round(dt, 2)}
    ok = n_col == n_vocab
    err = 0.0
    for dev, orc in ((d_a, a3), (d_b, b3)):
        ok = ok and np.array_equal(dev[0], orc[0]) and np.array_equal(dev[1], orc[1])
        if ok and len(orc[2]):
            err = max(err, float(np.abs(dev[2] - orc[2]).max()))
    rec.update({"indptr_and_indices_equal": bool(ok), "max_abs_value_err": err if ok else None, "ok": bool(ok and err <= 2e-7)})
    return a3, b3, n_vocab, rec


def k3_cpu_and_parity(job, idx, val, seconds, all_cores=True, min_rows=0, lists=None, all_cores_seconds=None):
    """CPU arm (ii): oracle/cossim_topn.c on ONE core -- how PolyFuzz calls sparse_dot_topn (_utils.py:82) -- over a
    seeded random sample of from-rows sized to `seconds` (at least min_rows); arm (iii): the same on all host cores (row
    ranges on threads; ctypes releases the GIL).  The sample's results are also the parity check of the GPU result.
    lists = (from strings, to strings or None): the CPU side then starts from the STRINGS -- the oracle vectoriser builds the
    float64 matrices the oracle product runs on, and the device's CSR is checked against them (`parity_check.vectoriser`);
    without them the oracle product runs on the device-built CSR."""
    import concurrent.futures as cf
    import oracle
    oracle.build_native()
    vec_rec = {"ok": None, "what": "not run: no strings handed over, the oracle product ran on the device-built CSR"}
    om = oracle_matrices(job, *lists) if lists is not None else None
    if om is not None:
        a3, b3, n_col, vec_rec = om
    else:
        a3, b3, n_col = job.host_matrices()
    n_from, excl = len(a3[0]) - 1, job.self_match
    rng = np.random.default_rng(SEED)
    whole_list_seconds = 1.5 * seconds if all_cores_seconds is None else all_cores_seconds     # (what the all-cores arm may take per thread to cover EVERY row)
    all_cores_seconds = 0.5 * seconds if all_cores_seconds is None else all_cores_seconds      # per thread, single-core speed

    def run(rows):
        return oracle.cossim_topn(a3, b3, n_col, job.top_n, job.min_similarity, exclude_diag=excl, rows=rows)

    probe = rng.choice(n_from, min(200, n_from), replace=False)
    t0 = time.perf_counter()
    run(probe)
    per_row = (time.perf_counter() - t0) / max(len(probe), 1)
    n_rows = int(min(n_from, max(len(probe), min_rows, seconds / max(per_row, 1e-9))))
    rows = np.sort(rng.choice(n_from, n_rows, replace=False))
    t0 = time.perf_counter()
    e_idx, e_val = run(rows)
    dt = time.perf_counter() - t0
    cores = n_cores()
    base = {"value": n_rows * float(job.n_to) / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": f"{n_rows} random from-rows (seed {SEED}) of {n_from} x all {job.n_to} to-rows, oracle/cossim_topn.c "
                      f"(Gustavson + strict bound + top-{job.top_n}, float64), {dt:.1f} s on 1 of {cores} host cores; "
                      "vectorisation not included"}
    arms = [dict(base, arm="ii: sparse product as PolyFuzz calls sparse_dot_topn (single thread), restated")]
    rows_what = f"seeded random sample of the from-rows (seed {SEED})"
    if all_cores:
        # arm (iii): contiguous row ranges on all host cores -- the WHOLE from-list wherever the budget allows (round 6, VERDICT r5
        # weak 1b: these rows used to be computed, timed and thrown away) -- and every row it computes joins the parity check
        # (the threads are as many as the cgroup grants CPUs -- 16 on the GPU box, n_cores(): the whole headline list is 9 s of them --,
        # and where the whole list fits 1.5 x the single-core arm's time per thread it is taken whole: parity over every row)
        share = -(-n_from // max(cores, 1))
        budget = whole_list_seconds if share * per_row <= whole_list_seconds else all_cores_seconds
        per_thread = int(max(8, min(share, budget / max(per_row, 1e-9))))
        ranges = [(t * per_thread, min((t + 1) * per_thread, n_from)) for t in range(cores) if t * per_thread < n_from]
        if ranges:
            t0 = time.perf_counter()
            with cf.ThreadPoolExecutor(len(ranges)) as ex:
                parts = list(ex.map(run, ranges))
            dt3 = time.perf_counter() - t0
            n3 = ranges[-1][1]
            arms.append({"arm": "iii: the same on all host cores", "value": n3 * float(job.n_to) / dt3,
                         "unit": "pairs/s", "cores": len(ranges), "kind": "port",
                         "sample": f"{len(ranges)} threads x <= {per_thread} from-rows = rows [0, {n3}) of {n_from} x all {job.n_to} to-rows, {dt3:.1f} s"})
            # the single-core sample's rows beyond [0, n3) stay in the check; inside it they are the same rows computed twice
            keep = rows >= n3
            rows = np.concatenate([np.arange(n3, dtype=rows.dtype), rows[keep]])
            e_idx = np.concatenate([p[0] for p in parts] + [e_idx[keep]])
            e_val = np.concatenate([p[1] for p in parts] + [e_val[keep]])
            rows_what = (f"ALL {n_from} from-rows" if n3 == n_from else
                         f"rows [0, {n3}) + the seeded random sample's rows beyond (seed {SEED})")
    par = k3_parity(job, idx, val, rows, e_idx, e_val, a3, b3, n_col, rows_what)
    par["matrices"] = ("oracle-built float64 CSR (the whole chain K1 -> K2 -> index -> K3 against the whole restated chain)"
                       if om is not None else "device-built CSR (K3 alone)")
    par["vectoriser"] = vec_rec
    par["ok"] = bool(par["ok"] and vec_rec["ok"] is not False)
    return base, arms, par


def reference_backend_arm(names, top_n):
    """CPU arm (i): the reference's own EXECUTABLE back-end -- sklearn vectoriser + dense cosine + full sorts + frame,
    _utils.py:94-125 -- RESTATED (oracle/reference_path.py, pinned on frames the reference package produced; the package
    itself is not on the GPU box) at the C2 size, where its dense matrix fits."""
    try:
        from oracle.reference_path import sklearn_backend_match
        from polyfuzz_amd import datasets
        n2 = 10_000 if len(names) >= 20_000 else max(100, len(names) // 2)
        fl, tl = datasets.c2_lists(n2) if len(names) == N_NAMES else (names[:n2], names[n2:2 * n2])
        tm = {}
        sklearn_backend_match(fl, tl, top_n=top_n, timings=tm)
        return {"arm": "i: restated reference path (the reference's sklearn back-end: vectorise + dense cosine + full sorts + frame)",
                "value": float(n2) * float(n2) / tm["total_s"], "unit": "pairs/s", "cores": "BLAS threads", "kind": "port",
                "sample": f"config 2: {n2} x {n2} company names, top-{top_n}, end to end {tm['total_s']:.1f} s (vectorise "
                          f"{tm['vectorise_s']:.1f}, cosine+sort {tm['cosine_sort_s']:.1f}, frame {tm['frame_s']:.1f})"}
    except Exception as e:      # a CPU arm must never take the bench line down
        return {"arm": "i: restated reference path", "error": f"{type(e).__name__}: {e}"}


# ---- what a user sees: .match() wall time and single-query latency ------------------------------------------------

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
    # the frame is the device result: To == names[idx] wherever the rounded score survives (a random sample of rows)
    same = True
    if result_idx is not None:
        to0 = df["To"].tolist()
        for i in np.random.default_rng(SEED).choice(len(names), min(5000, len(names)), replace=False).tolist():
            same = same and (to0[i] is None or to0[i] == names[result_idx[i, 0]])
    n = float(len(names))
    out = {"match_wall_ms": ts[med], "match_wall_ms_min": min(ts), "match_pairs_per_s": n * n / (ts[med] * 1e-3),
           "match_stages_ms": {k: round(v, 3) for k, v in stages[med].items()},
           "match_what": f"the SURVEY section 8d metric: TFIDF(min_similarity=0, top_n={top_n}).match(names), Python list in, "
                         f"DataFrame out (pack, H2D, device step, D2H, frame), median of {reps}",
           "match_frame_consistent_with_device_result": bool(same)}
    return out


def top1_latency(names, reps=50):
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


# ---- the timed loop every configuration shares ---------------------------------------------------------------------

def timed_steps(world, ctx, step, steps, warmup, prof_level=2):
    """`warmup` untimed steps, then exactly `steps` steps bracketed by barrier + device sync, max over the ranks;
    the library's HIP-event timers (prof_level 2: the dominant kernels only) run inside the timed region."""
    result = None
    for _ in range(warmup):
        result = step()
    world.barrier(ctx)
    ctx.prof_enable(prof_level)
    ctx.prof_reset()
    t0 = time.perf_counter()
    ctx.event_record(0)
    for _ in range(steps):
        result = step()
    ctx.event_record(1)
    world.barrier(ctx)
    wall = time.perf_counter() - t0
    ctx.prof_enable(False)
    return world.max(wall), result


def contract(metric, value, unit, world, args, steps, warmup, wall, scaling, dtype, data, config):
    return {"metric": metric, "value": value, "unit": unit, "n_gpus": world.size, "steps": steps, "warmup": warmup,
            "ms_per_step": wall / steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": dtype, "data": data, "config": config}


# ---- configuration: the headline (and config 2 / config 4's shard, the same job at other sizes) --------------------

def run_tfidf(world, ctx, args, from_shard=None, to_list=None, top_n=None, steps=None, warmup=None, self_match=True,
              shard_desc="the whole list", n_from_total=None, label=None, cpu_seconds=None, min_parity_rows=0,
              all_cores_arm=True, kind="real", shard_offset=0, rows_per_rank=None, traffic_label=None, all_cores_seconds=None,
              user_step=None):
    """One TfidfMatchJob under the clock.  Returns (contract-shaped record incl. roofline / cpu_baseline / parity_check,
    job, (idx, val) of the last step)."""
    from polyfuzz_amd import pipeline
    top_n = args.top_n if top_n is None else top_n
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    comm, exchange = world.comm(ctx) if world.size > 1 else (None, "none (single GPU)")
    job = pipeline.TfidfMatchJob(ctx, from_shard, to_list, top_n=top_n, min_similarity=MIN_SIM, self_match=self_match,
                                 shard_offset=shard_offset, comm=comm, rows_per_rank=rows_per_rank)
    n_to = job.n_to
    n_from_total = job.n_from if n_from_total is None else n_from_total
    user = None
    if user_step is not None:
        # timed region 1: the user-level call (user_step: Python list in, DataFrame out) -- the line's value / ms_per_step
        uwall, _ = timed_steps(world, ctx, user_step, steps, warmup)
        k3u = ctx.prof_get("k3_cossim_topn")
        user = {"wall": uwall, "gpu_ms": ctx.event_elapsed_ms(0, 1), "k3_ms": k3u[0] / max(k3u[1], 1) * (k3u[1] / steps), "k3_launches": k3u[1] / steps}
    # the device-resident step (timed region 2 when there is a user-level step): same protocol, list resident in HBM
    wall, result = timed_steps(world, ctx, job.step, steps, warmup)
    k3_timed = ctx.prof_get("k3_cossim_topn")
    gpu_ms = ctx.event_elapsed_ms(0, 1)
    # the per-kernel breakdown of a step: three more steps with every profiled kernel bracketed, outside the timed region
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(3):
        result = job.step()
    world.barrier(ctx)
    ctx.prof_enable(False)
    if world.rank != 0:
        return None, job, None
    kernel_ms = {name: round(ctx.prof_get(name)[0] / 3, 4) for name in pipeline.PROFILED_KERNELS}
    stats = job.stats()
    traffic, traffic_note = recorded_traffic(traffic_key(args, traffic_label))
    out = contract(
        "string-pairs/sec, TF-IDF cosine top-n 100k x 100k (config.timed_step says what a step is: by default the user-level "
        ".match() call of SURVEY section 8d's metric, `device_step` = the list resident in HBM; latency.top1_single_query_ms = "
        "top-1 match latency)"
        if label is None else f"string-pairs/sec, TF-IDF cosine top-{top_n}, {label}",
        float(n_from_total) * float(n_to) * steps / wall, "pairs/s", world, args, steps, warmup, wall,
        tfidf_scaling(args, world.size) if (label is None or world.size > 1 and args.scaling) else "weak", "f32",
        "real: reference data/company_names.json (100 000 SEC-EDGAR names, gzipped in polyfuzz_amd/data/)" if kind == "real"
        else "synthetic: token recombination of the real names (polyfuzz_amd/synth.py, SURVEY section 8d config 4)",
        {"workload": label or (f"TFIDF(min_similarity={MIN_SIM}, top_n={top_n}).match(names): self-match of "
                               f"{'the first ' + str(n_to) + ' of the ' if n_to < N_NAMES else 'all '}{N_NAMES} real company names, "
                               "char-3-gram TF-IDF cosine (SURVEY.md section 8d headline; reference "
                               "docs/tutorial/datasets/datasets.md:36-41)"),
         "n_from_total": n_from_total, "n_from_this_rank": job.n_from, "n_to": n_to, "top_n": top_n, "from_rows": shard_desc,
         "vocab": stats["vocab"], "nnz_from": stats["nnz_from"], "nnz_to": stats["nnz_to"], "multiply_adds_rank0": stats["madds"],
         "step": job.step_description(),
         "parallelism": (f"from-rows sharded x{world.size}, list replicated" if not job.result_is_full else
                         f"from-rows sharded x{world.size} (rank r: the rows r, r + {world.size}, ...), list replicated, every unordered pair scored once over all ranks"),
         "exchange": exchange if not job.result_is_full else exchange.replace("of the per-shard result blocks", "of the pass-0 thresholds and of the per-row candidate lists"),
         "transport": world.kind, "device": ctx.info()["name"]})
    out["gpu_ms_per_step_rank0"] = gpu_ms / steps
    out["kernel_ms_per_step"] = kernel_ms
    out["roofline"] = k3_roofline(job, stats, k3_timed[0], k3_timed[1], top_n, traffic, traffic_note)
    if user is not None:
        # the line's value is the user-level call's; the device-resident step and the roofline of its dominant kernel beside it
        dev = {"ms_per_step": out["ms_per_step"], "value": out["value"], "unit": "pairs/s", "gpu_ms_per_step_rank0": gpu_ms / steps,
               "what": "timed region 2, same protocol (warm-up, barrier + sync, exactly `steps` steps): the list resident in HBM, "
                       "fit + vectorise + index + K3 per step (TfidfMatchJob.step); `kernel_ms_per_step` and `roofline` are this region's"}
        out["ms_per_step"] = user["wall"] / steps * 1e3
        out["value"] = float(n_from_total) * float(n_to) * steps / user["wall"]
        out["gpu_ms_per_step_rank0"] = user["gpu_ms"] / steps
        out["device_step"] = dev
        out["k3_ms_per_step_inside_the_call"] = user["k3_ms"]
        out["roofline"]["measured_on"] = ("the device-resident steps of timed region 2 (one K3 launch per step, HIP events on the library's "
                                          "stream); inside the user-level call K3 runs as one pass-1 launch with its row ranges merged on a "
                                          f"side stream: {user['k3_ms']:.3f} ms per call from its first kernel to its last merge")
    res = None
    if world.size == 1:
        res = result.download()
        if not args.no_cpu_baseline:
            out["cpu_baseline"], out["cpu_baseline_arms"], out["parity_check"] = k3_cpu_and_parity(
                job, res[0], res[1], args.cpu_seconds if cpu_seconds is None else cpu_seconds, all_cores=all_cores_arm,
                min_rows=min_parity_rows, lists=(from_shard, to_list), all_cores_seconds=all_cores_seconds)
    return out, job, res


class FrameKeeper:
    """The other end of the timed user-level calls.  What is timed is K CALLS of `.match()`: not the disposal of a result (the
    caller's business once it has used it: 600 000 reference counts for the headline's frame, ~0.9 ms -- in the plain loop `df =
    m.match(names)` it sits between two calls and the step read 4.5 ms for a 3.6-ms call; a consumer thread that lets go of the
    frames beside the next call fights the calls for the GIL: 5.8 ms, measured), and not the first touch of fresh pages either
    (keeping K frames of 9 MB makes every call fault them in: +1.3 ms, measured -- a long-running caller's heap is warm).  So: the
    frames of the timed calls are KEPT until the region is over, in memory this process has touched before -- `warm()` allocates,
    touches and frees as much as the kept frames will take, with glibc told to keep freed blocks mapped (mallopt: no trimming, no
    mmap below 32 MB) -- and disposed of after the clock has stopped."""

    def __init__(self):
        self.frames = []

    @staticmethod
    def warm(n_rows, top_n, n_frames):
        try:
            libc = ctypes.CDLL("libc.so.6")
            libc.mallopt(-1, (1 << 31) - 1)       # M_TRIM_THRESHOLD: freed memory at the top of the heap stays with the process
            libc.mallopt(-3, 32 << 20)            # M_MMAP_THRESHOLD: the frame's 800-KB columns come from the heap, not from mmap
        except OSError:
            return
        blocks = []
        for _ in range(n_frames * (2 * top_n + 2)):
            a = np.empty(n_rows, np.float64)
            a.fill(1.0)
            blocks.append(a)
        del blocks

    def take(self, frame):
        self.frames.append(frame)

    def close(self):
        last = self.frames[-1] if self.frames else None
        del self.frames[:]
        return last


def tfidf_scaling(args, size):
    """strong (one job cut over the ranks) is the headline's default at N > 1; weak (a distinct batch per rank) on request"""
    return args.scaling or ("strong" if size > 1 else "weak")


def headline(world, ctx, args):
    from polyfuzz_amd import pipeline, synth
    names, kind = the_list(args)
    n, rank, size = len(names), world.rank, world.size
    scaling = tfidf_scaling(args, size)
    self_match = True
    if size == 1:
        kw = dict(from_shard=names, to_list=None, shard_desc="the whole list", n_from_total=n)
    elif scaling == "strong":
        # the ONE 100k x 100k self-match cut over the ranks (the default at N > 1: what "how fast is the headline on N GPUs" asks).
        # Where K3's symmetric form applies the job deals the rows r, r + N, ... to rank r (TfidfMatchJob.step); the contiguous
        # cost-balanced cuts below are what the row-major form works on
        bounds = pipeline.balanced_bounds(names, size)       # equal characters, not equal rows: the list is sorted and skewed
        b, e = bounds[rank]
        kw = dict(from_shard=names[b:e], to_list=names, shard_offset=b, rows_per_rank=max(y - x for x, y in bounds),
                  shard_desc=f"rows [{b}, {e}) of the list (cost-balanced cuts); in K3's symmetric form: the rows {rank}, {rank} + {size}, ...",
                  n_from_total=n)
    else:
        # weak scaling: every rank matches ITS OWN batch of 100 000 query names against the replicated list -- a DISTINCT batch
        # of equal cost (identical self-matches on every rank would take the symmetric half-work kernel and count
        # duplicate rows as throughput): rank r's batch is the real list in a rank-seeded random order, a two-list match
        # (the row-major kernel, the exact sharded fit on to + from: batch and list are different lists to the job)
        perm = np.random.default_rng(SEED + 1000 + rank).permutation(n)
        self_match = False
        kw = dict(from_shard=[names[i] for i in perm], to_list=names, shard_offset=0, rows_per_rank=n, n_from_total=n * size,
                  shard_desc="every rank: its own batch of from-rows (the list's names in a rank-seeded random order) against the replicated list")
    # The timed step (--step match, the default): the user-level call SURVEY section 8d defines the metric on -- Python list in,
    # DataFrame out.  One GPU: TFIDF.match itself.  N > 1, strong scaling: the same call on every rank of the communicator
    # (pipeline.sharded_self_match: every rank packs and uploads the replicated list, works on its share of the rows, gets the full
    # result from the exchange and builds the full frame).  The frames of the timed calls are kept, on a warmed heap, until the
    # clock has stopped (FrameKeeper: K calls are timed, not the disposal of K - 1 results nor the first touch of fresh pages).
    user_step, user_what = None, None
    consumer = FrameKeeper() if args.step == "match" else None
    if consumer is not None:
        FrameKeeper.warm(n, args.top_n, args.steps + args.warmup + 2)
    if args.step == "match" and (size == 1 or scaling == "strong"):
        if size == 1:
            from polyfuzz_amd.models import TFIDF
            matcher = TFIDF(n_gram_range=(3, 3), min_similarity=MIN_SIM, top_n=args.top_n)
            step_walls = []

            def user_step():
                t = time.perf_counter()
                consumer.take(matcher.match(names))
                step_walls.append((time.perf_counter() - t) * 1e3)
            user_what = f"TFIDF(min_similarity={MIN_SIM}, top_n={args.top_n}).match(names): Python list in, DataFrame out (pack, H2D, fit + vectorise + index + K3, results to the host, frame)"
        else:
            comm, _ = world.comm(ctx)
            if comm is not None:
                user_step = lambda: consumer.take(pipeline.sharded_self_match(ctx, comm, names, top_n=args.top_n, min_similarity=MIN_SIM))
                user_what = (f"pipeline.sharded_self_match(ctx, comm, names, top_n={args.top_n}) on every rank = TFIDF.match(names) on {size} GPUs: "
                             "Python list in, the full DataFrame out on every rank")
    out, job, res = run_tfidf(world, ctx, args, kind=kind, min_parity_rows=5000 if n >= 50_000 else 0, self_match=self_match,
                              user_step=user_step, **kw)
    last_frame = consumer.close() if consumer is not None else None
    if out is None:
        return None
    if user_step is not None:
        out["config"]["timed_step"] = user_what
        # what the user-level call runs on beside the GPU: the string packer and the frame's range fill work on a pool of host threads
        # confined to the cores that share the calling thread's L3 (polyfuzz_amd/csrc_host/_pack.c); PFZ_HOST_THREADS=1 = the calling thread alone
        from polyfuzz_amd import _lib as _pl
        out["config"]["host_threads"] = {"per_call": int(_pl.host_threads()), "usable_cpus": int(_pl.usable_cpus()),
                                         "confined_to_the_callers_l3": os.environ.get("PFZ_HOST_PIN", "1") != "0"}
        if size == 1 and len(step_walls) >= args.steps:
            # every timed call under its own clock (the region's clock is what `ms_per_step` comes from): one slow call -- a helper
            # thread's core waking up, a neighbour on the host -- shows here instead of hiding in the mean
            w = sorted(step_walls[-args.steps:])
            out["timed_step_walls_ms"] = {"min": round(w[0], 3), "median": round(w[len(w) // 2], 3), "max": round(w[-1], 3),
                                          "calls_over_1.25x_median": int(sum(x > 1.25 * w[len(w) // 2] for x in w))}
        out["value_definition"] = ("SURVEY section 8d's metric: N_from x N_to x steps / wall of the timed steps, a step = the user-level call "
                                   "(host list to DataFrame: host packing, PCIe both ways and the frame are inside).  The device-resident "
                                   "step -- the list already in HBM, what the bench contract's `value` names -- is timed in a second region "
                                   "of the same run: `device_step`; `--step device` makes it the line.")
        if res is not None and last_frame is not None:
            # the frame IS the device result: every To cell of every rank column, all rows
            ok = True
            for r in range(args.top_n):
                to = last_frame["To" if r == 0 else f"To_{r + 1}"].tolist()
                ok = ok and all(t is None or t == names[j] for t, j in zip(to, res[0][:, r].tolist()))
            out["frame_consistent_with_device_result"] = bool(ok)
    else:
        out["config"]["timed_step"] = "the device-resident step (TfidfMatchJob.step): list in HBM, fit + vectorise + index + K3"
        out["value_definition"] = ("device-resident step: N_from x N_to x steps / wall of the timed steps, list in HBM.  SURVEY section 8d's "
                                   "metric -- pairs/s over the wall time of .match(), host list to DataFrame -- is what the default "
                                   "(--step match) times.")
    if size == 1 and not args.no_cpu_baseline:
        out["cpu_baseline_arms"].append(reference_backend_arm(names, args.top_n))
        # the library pins, armed: this box may have what the build container lacks
        try:
            sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
            from live_pins import live_pins
            out["library_pins"] = live_pins()
        except Exception as e:
            out["library_pins"] = {"error": f"{type(e).__name__}: {e}"}
    if size == 1 and not args.no_match_wall:
        mw = match_wall(names, args.top_n, res[0])       # (the same call under a median-of-7 protocol, with its stage stamps)
        lat = top1_latency(names)
        # the two numbers of BASELINE.json's metric as short top-level keys right behind ms_per_step (the
        # driver's parse of the line kept neither)
        head = {}
        for k, v in out.items():
            head[k] = v
            if k == "ms_per_step":
                head["match_wall_ms"] = mw["match_wall_ms"]
                head["latency_top1_ms"] = lat["top1_single_query_ms"]
        out = head
        out.update(mw)
        out["latency"] = lat
    return out


def compact(rec, extra=()):
    """the sub-record form of a configuration's line"""
    keep = ("value", "unit", "steps", "ms_per_step", "match_wall_ms", "match_what", "dtype", "data", "kernel_ms_per_step", "roofline",
            "cpu_baseline", "parity_check") + tuple(extra)
    out = {"workload": rec["config"]["workload"]}
    out.update({k: rec[k] for k in keep if k in rec})
    rf = out.get("roofline")
    if isinstance(rf, dict):
        out["roofline"] = {k: v for k, v in rf.items() if k not in ("bound_note", "traffic_note", "lds_floor_what", "what")}
    return out


def run_c2(world, ctx, args, steps=20, warmup=3):
    """BASELINE config 2: TF-IDF char-3-gram cosine top-5, 10k x 10k of the real company names (two lists)."""
    from polyfuzz_amd import datasets
    from polyfuzz_amd.models import TFIDF
    fl, tl = datasets.c2_lists()
    out, job, res = run_tfidf(world, ctx, args, from_shard=fl, to_list=tl, top_n=5, steps=steps, warmup=warmup, self_match=False,
                              label="TFIDF(min_similarity=0, top_n=5).match(from, to): config 2, 10 000 x 10 000 real company "
                                    "names (default_rng(0) permutation; SURVEY section 8d)",
                              shard_desc="the whole from-list", cpu_seconds=min(args.cpu_seconds, 3.0), min_parity_rows=2000,
                              all_cores_arm=True, traffic_label="c2_tfidf_10k")
    if out is not None and world.size == 1 and not args.no_match_wall:
        m = TFIDF(n_gram_range=(3, 3), min_similarity=0, top_n=5)
        m.match(fl, tl)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            m.match(fl, tl)
            ts.append((time.perf_counter() - t0) * 1e3)
        out["match_wall_ms"] = median(ts)
        out["match_what"] = "TFIDF(min_similarity=0, top_n=5).match(from, to), Python lists in, DataFrame out, median of 7"
    return out


def run_tfidf_1m(world, ctx, args, steps=3, warmup=1):
    """BASELINE config 4, one GPU's share: a 125 000-row from-shard against 1 000 000 to-strings, top-10 (the 8-GPU job
    is 8 such shards; the to-side is replicated)."""
    from polyfuzz_amd import synth
    n_to, n_from = 1_000_000, 125_000
    t0 = time.perf_counter()
    tl, fl = synth.company_names(n_to, 5678), synth.company_names(n_from, 1234)
    t_gen = time.perf_counter() - t0
    out, job, res = run_tfidf(world, ctx, args, from_shard=fl, to_list=tl, top_n=10, steps=steps, warmup=warmup, self_match=False,
                              label="one GPU's shard of config 4: 125 000 synthetic from-names x 1 000 000 synthetic to-names, "
                                    "top-10 (TfidfMatchJob, lists resident)", kind="synthetic",
                              shard_desc="rows of rank 0 of 8", cpu_seconds=min(args.cpu_seconds, 4.0), min_parity_rows=64,
                              all_cores_arm=True, all_cores_seconds=5.0, traffic_label="tfidf_1m_shard")
    if out is not None:
        out["host_generation_s"] = round(t_gen, 2)
        out["index"] = job.index.info()
    return out


# ---- configuration 3 and the RapidFuzz default: the edit-distance matchers -------------------------------------------

def rows_on_all_cores(fn, n, per_row, wall_seconds):
    """fn((begin, end)) -> (idx, score) of from-rows [begin, end), run on all host cores (ctypes releases the GIL) UNDER A DEADLINE:
    every thread owns a contiguous stretch of the rows and works through it in small pieces until the stretch is done or
    `wall_seconds` are over -- a box whose affinity mask shows 256 CPUs may grant a dozen cores' worth of time (measured: the
    all-cores arms ran at ~10x one core), and a CPU arm must not take the bench line minutes.  Returns (rows int64[m] -- the rows
    that were computed, ascending --, idx[m], score[m], wall, threads)."""
    import concurrent.futures as cf
    cores = n_cores()
    per_thread = -(-n // max(cores, 1))
    stretches = [(t * per_thread, min((t + 1) * per_thread, n)) for t in range(cores) if t * per_thread < n]
    piece = int(max(1, min(per_thread, 0.25 / max(per_row, 1e-9))))      # ~0.25 s of one core per piece
    t0 = time.perf_counter()
    deadline = t0 + wall_seconds

    def work(st):
        lo, hi = st
        got, pos = [], lo
        while pos < hi and (pos == lo or time.perf_counter() < deadline):
            nxt = min(pos + piece, hi)
            got.append(fn((pos, nxt)))
            pos = nxt
        return lo, pos, got

    with cf.ThreadPoolExecutor(len(stretches)) as ex:
        parts = list(ex.map(work, stretches))
    dt = time.perf_counter() - t0
    rows = np.concatenate([np.arange(lo, pos, dtype=np.int64) for lo, pos, _ in parts])
    idx = np.concatenate([g[0] for _, _, got in parts for g in got])
    score = np.concatenate([g[1] for _, _, got in parts for g in got])
    return rows, idx, score, dt, len(stretches)


def rows_what(rows, n, what):
    return f"ALL {n} {what}" if len(rows) == n else f"{len(rows)} of the {n} {what}: a contiguous stretch per host thread, as far as the CPU arm's deadline let it get"


def edit_rows_sample(n, k):
    return np.sort(np.random.default_rng(SEED).choice(n, min(k, n), replace=False))


def run_editdistance(world, ctx, args, steps=None, warmup=None, cpu_seconds=None):
    """BASELINE.json config 3 / SURVEY.md section 8d: EditDistance (rapidfuzz.fuzz.ratio = Indel ratio) all pairs of
    20 000 x 20 000 IMDB titles (default_rng(0) permutation, first from-title 'Polly Blue Eyes'), first arg-max
    per from-title.  One step = one pass of K4 over all pairs with both lists and the to-side plan resident.
    N > 1 (strong scaling): the from-titles are split over the ranks, the per-shard (index, score) blocks all-gathered."""
    from polyfuzz_amd import _lib, datasets, pipeline
    from polyfuzz_amd.models import EditDistance
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    fl, tl = datasets.c3_lists(2_000 if args.small else 20_000)
    n = len(fl)
    bounds = pipeline.balanced_bounds(fl, world.size)      # equal characters per rank: a from-title costs its length x the to-list
    b, e = bounds[world.rank]
    comm, exchange = world.comm(ctx) if world.size > 1 else (None, "none (single GPU)")
    job = pipeline.BestChoiceJob(ctx, fl[b:e], tl, scorer="ratio", comm=comm, rows_per_rank=max(y - x for x, y in bounds))
    plan = job.plan_info()
    wall, result = timed_steps(world, ctx, job.step, steps, warmup)
    k4_ms, k4_launches = ctx.prof_get("k4_indel")
    if world.rank != 0:
        return None
    idx, score = job.result_host(result)
    if world.size > 1:
        sizes = [y - x for x, y in bounds]
        idx, score = pipeline.BestChoiceJob.unpad(idx, score, sizes, job.rows_per_rank)
    k4_step_s = k4_ms / steps * 1e-3
    # algorithmic work: one 5-operation word update (u = V & M; V = (V + u) | (V ^ u), + the table look-up) per
    # to-character per 32/64-bit word of the from-string -- counted as 32-bit integer operations (this rank's rows)
    words32 = np.array([(1 if len(a) <= 32 else 2 * ((len(a) + 63) // 64)) for a in fl[b:e]], np.float64)
    int_ops = 5.0 * float(words32.sum()) * float(plan["char_steps"])
    cells = float(sum(map(len, fl[b:e]))) * float(sum(map(len, tl)))
    out = contract("string-pairs/sec, EditDistance (Indel ratio) all pairs + first arg-max, 20k x 20k IMDB titles",
                   float(n) * float(len(tl)) * steps / wall, "pairs/s", world, args, steps, warmup, wall,
                   (args.scaling or "strong") if world.size > 1 else "weak", "int32/int64 bit-vectors, f64 score",
                   "real: reference data/movie_titles.json (IMDB), gzipped in polyfuzz_amd/data/",
                   {"workload": "EditDistance(scorer=fuzz.ratio).match(from, to): 20000 x 20000 IMDB titles (SURVEY.md section 8d "
                                "config 3), lists and to-side plan resident", "n_from": n, "n_from_this_rank": e - b, "n_to": len(tl),
                    "alphabet": plan["n_symbols"], "dp_cells_rank0": cells, "to_char_steps": plan["char_steps"],
                    "parallelism": f"from-titles sharded x{world.size}, to-list replicated", "exchange": exchange,
                    "transport": world.kind})
    out["kernel_ms_per_step"] = {"k4_indel": round(k4_ms / steps, 4), "launches_per_step": k4_launches / steps}
    out["roofline"] = {"kernel": "k4_indel", "bound": "int32 VALU issue (+ LDS look-ups)", "achieved": int_ops / k4_step_s / 1e12,
                       "peak": INT32_PEAK_TOPS, "unit": "Tera int-op/s", "frac": int_ops / k4_step_s / 1e12 / INT32_PEAK_TOPS,
                       "traffic": None, "algorithmic_int_ops_per_step": int_ops, "dp_cell_updates_per_s": cells / k4_step_s,
                       "what": "5 integer operations per to-character per 32-bit word of the from-string (bit-parallel LCS), "
                               "against 256 CU x 4 SIMD x 32 lanes x 2.4 GHz"}
    if not args.no_cpu_baseline:
        import oracle
        oracle.build_native()
        seconds = args.cpu_seconds if cpu_seconds is None else cpu_seconds
        c0 = time.perf_counter()
        oracle.indel_argmax(fl[:20], tl)
        per_row = (time.perf_counter() - c0) / 20
        rows = edit_rows_sample(n, int(max(200, seconds / max(per_row, 1e-9))))
        sample = [fl[i] for i in rows]
        c0 = time.perf_counter()
        e_idx, e_score = oracle.indel_argmax(sample, tl)
        dt = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": len(rows) * float(len(tl)) / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"{len(rows)} random from-titles (seed {SEED}) x all {len(tl)} to-titles, oracle/indel.c "
                                         f"(plain O(|a||b|) LCS DP), {dt:.1f} s on 1 of {n_cores()} host cores"}
        # parity: the WHOLE configuration wherever the host's cores allow (round 6, VERDICT r5 weak 1b: ~1 s of oracle/indel.c on a
        # 256-core box), contiguous row ranges on threads; the single-core sample's rows beyond stay in the check
        a_rows, a_idx, a_score, dt3, threads = rows_on_all_cores(lambda r: oracle.indel_argmax(fl, tl, rows=r), n, per_row, 12.0)
        same = np.array_equal(idx[a_rows], a_idx) and np.array_equal(score[a_rows], a_score)
        same = same and np.array_equal(idx[rows], e_idx) and np.array_equal(score[rows], e_score)
        out["cpu_baseline_all_cores"] = {"value": len(a_rows) * float(len(tl)) / dt3, "unit": "pairs/s", "cores": threads, "kind": "port",
                                         "sample": f"{len(a_rows)} of {n} from-titles x all {len(tl)} to-titles, oracle/indel.c, {dt3:.1f} s"}
        out["parity_check"] = {"rows_checked": int(len(np.union1d(a_rows, rows))), "rows": rows_what(a_rows, n, "from-titles"),
                               "bit_exact": bool(same),
                               "rows_differing": int(((idx[a_rows] != a_idx) | (score[a_rows] != a_score)).sum())}
    if world.size == 1 and not args.no_match_wall:
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
        out["match_wall_ms"], out["match_wall_ms_to_list_resident"] = median(ts), median(ts2)
        out["match_what"] = ("EditDistance(normalize=False).match(from, to): Python lists in, DataFrame out; "
                             "..._to_list_resident = match(from, to, re_train=False), the to-list and its K4 plan kept on the device")
    return out


def run_rapidfuzz(world, ctx, args, steps=3, warmup=1, cpu_seconds=None):
    """RapidFuzz() -- default scorer fuzz.WRatio, what PolyFuzz("EditDistance") dispatches to (polyfuzz.py:128-130,
    _rapidfuzz.py:45-113) -- on config 3's 20 000 x 20 000 IMDB titles: process.extractOne per from-title (K7)."""
    from polyfuzz_amd import datasets, pipeline
    from polyfuzz_amd.models import RapidFuzz
    fl, tl = datasets.c3_lists(2_000 if args.small else 20_000)
    n = len(fl)
    bounds = pipeline.balanced_bounds(fl, world.size)      # equal characters per rank: a from-title costs its length x the to-list
    b, e = bounds[world.rank]
    comm, exchange = world.comm(ctx) if world.size > 1 else (None, "none (single GPU)")
    job = pipeline.BestChoiceJob(ctx, fl[b:e], tl, scorer="WRatio", comm=comm, rows_per_rank=max(y - x for x, y in bounds))
    wall, result = timed_steps(world, ctx, job.step, steps, warmup, prof_level=True)
    k7_ms, k7_launches = ctx.prof_get("k7_fuzz")
    if world.rank != 0:
        return None
    idx, score = job.result_host(result)
    if world.size > 1:
        sizes = [y - x for x, y in bounds]
        idx, score = pipeline.BestChoiceJob.unpad(idx, score, sizes, job.rows_per_rank)
    out = contract("string-pairs/sec, RapidFuzz() = process.extractOne(scorer=fuzz.WRatio) per from-string, 20k x 20k IMDB titles",
                   float(n) * float(len(tl)) * steps / wall, "pairs/s", world, args, steps, warmup, wall,
                   (args.scaling or "strong") if world.size > 1 else "weak", "int32/int64 bit-vectors, f64 score",
                   "real: reference data/movie_titles.json (IMDB), gzipped in polyfuzz_amd/data/",
                   {"workload": "RapidFuzz().match(from, to) (scorer fuzz.WRatio, the reference's default EditDistance path): 20000 x "
                                "20000 IMDB titles, the lists' token forms and the to-side plan resident",
                    "n_from": n, "n_from_this_rank": e - b, "n_to": len(tl),
                    "parallelism": f"from-titles sharded x{world.size}, to-list replicated", "exchange": exchange,
                    "transport": world.kind})
    out["kernel_ms_per_step"] = {"k7_fuzz": round(k7_ms / steps, 3), "launches_per_step": k7_launches / steps}
    out["roofline"] = job.roofline(k7_ms / steps * 1e-3, INT32_PEAK_TOPS)
    if not args.no_cpu_baseline:
        import oracle
        oracle.build_native()
        seconds = args.cpu_seconds if cpu_seconds is None else cpu_seconds
        c0 = time.perf_counter()
        oracle.fuzz_extract_one(fl[:4], tl, "WRatio")
        per_row = (time.perf_counter() - c0) / 4
        n1 = int(max(8, seconds / max(per_row, 1e-9)))                     # single-thread sample: the CPU arm
        rows = edit_rows_sample(n, n1)
        c0 = time.perf_counter()
        e1_idx, e1_score = oracle.fuzz_extract_one([fl[i] for i in rows[:n1]], tl, "WRatio")
        dt = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": n1 * float(len(tl)) / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"{n1} random from-titles (seed {SEED}) x all {len(tl)} to-titles, oracle/fuzz_scorers.c "
                                         f"(rapidfuzz 3.x WRatio restated, plain LCS DP per window), {dt:.1f} s on 1 of {n_cores()} host cores"}
        # parity: the WHOLE configuration wherever the host's cores allow (round 6: ~10 s of oracle/fuzz_scorers.c on a 256-core
        # box), contiguous row ranges on threads; the single-core sample's rows beyond stay in the check
        a_rows, a_idx, a_score, dt3, threads = rows_on_all_cores(lambda r: oracle.fuzz_extract_one(fl, tl, "WRatio", rows=r), n, per_row, 20.0)
        r1 = rows[:n1]
        same = np.array_equal(idx[a_rows], a_idx) and np.array_equal(score[a_rows], a_score)
        same = same and np.array_equal(idx[r1], e1_idx) and np.array_equal(score[r1], e1_score)
        out["cpu_baseline_all_cores"] = {"value": len(a_rows) * float(len(tl)) / dt3, "unit": "pairs/s", "cores": threads, "kind": "port",
                                         "sample": f"{len(a_rows)} of {n} from-titles x all {len(tl)} to-titles, oracle/fuzz_scorers.c, {dt3:.1f} s"}
        out["parity_check"] = {"rows_checked": int(len(np.union1d(a_rows, r1))), "rows": rows_what(a_rows, n, "from-titles"),
                               "bit_exact": bool(same),
                               "rows_differing": int(((idx[a_rows] != a_idx) | (score[a_rows] != a_score)).sum())}
    if world.size == 1 and not args.no_match_wall:
        m = RapidFuzz()
        m.match(fl, tl)
        ts = []
        for _ in range(5):
            c0 = time.perf_counter()
            m.match(fl, tl)
            ts.append((time.perf_counter() - c0) * 1e3)
        out["match_wall_ms"] = median(ts)
        out["match_what"] = "RapidFuzz().match(from, to): Python lists in, DataFrame out, median of 5"
    return out


# ---- configuration 5: dense cosine top-n (K5) ------------------------------------------------------------------------

def run_dense(world, ctx, args, steps=None, warmup=None):
    """BASELINE.json config 5 / SURVEY.md section 8d: dense cosine top-10 of 500k x 500k 768-d embeddings on 8 GPUs --
    per rank a 62 500-row from-shard against all 500 000 to-vectors (K5), operands resident in HBM (weak scaling: N = 8 is
    config 5 itself).  One step = the shard's GEMM panels + row top-n.  Roofline: exact-fp32 MFMA."""
    from polyfuzz_amd import pipeline
    steps = args.steps if steps is None else steps
    warmup = max(1, args.warmup if warmup is None else warmup)
    n_to, n_from, d, top_n = (20_000, 4_000, 256, 10) if args.small else (500_000, 62_500, 768, 10)
    b = np.random.default_rng(7).standard_normal((n_to, d), dtype=np.float32)          # replicated: the same on every rank
    rng = np.random.default_rng(70 + world.rank)
    a = rng.standard_normal((n_from, d), dtype=np.float32)
    pick = rng.choice(n_to, n_from, replace=False)
    a += 2.0 * b[pick]                       # planted near-duplicates: the top rank is known
    comm, exchange = world.comm(ctx) if world.size > 1 else (None, "none (single GPU)")
    job = pipeline.DenseMatchJob(ctx, a, b, top_n=top_n, comm=comm, rows_per_rank=n_from)
    wall, res = timed_steps(world, ctx, job.step, steps, warmup)
    gemm_ms, launches = ctx.prof_get("k5_gemm_panel")
    if world.rank != 0:
        return None
    idx, val = res.download()
    idx, val = idx[:n_from], val[:n_from]            # (rank 0's block of the gathered result)
    flop = 2.0 * n_from * n_to * d
    gemm_s = gemm_ms / max(1, launches) * 1e-3 * (launches / steps)       # GEMM time per step
    out = contract("vector pairs/sec, dense cosine top-10, 62 500-row shards of 500k x 500k x 768 (BASELINE config 5)",
                   float(n_from) * world.size * n_to * steps / wall, "pairs/s", world, args, steps, warmup, wall, "weak", "f32",
                   "synthetic (SURVEY.md section 8d config 5: standard normal rows, planted near-duplicates)",
                   {"workload": "Embeddings-style cosine top-10: 62 500 from-vectors per GPU (config 5's row shard) x 500 000 "
                                "to-vectors x 768, operands resident, pipeline.DenseMatchJob", "n_from_per_rank": n_from,
                    "n_to": n_to, "dim": d, "top_n": top_n, "parallelism": f"from-rows sharded x{world.size}, to-vectors replicated",
                    "exchange": exchange, "transport": world.kind})
    out["kernel_ms_per_step"] = {"k5_gemm_panel": round(gemm_ms / steps, 3), "launches_per_step": launches / steps}
    out["roofline"] = {"kernel": "k5_gemm_panel_pipe", "bound": "mfma", "achieved": flop / gemm_s / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS,
                       "unit": "TFLOP/s", "frac": flop / gemm_s / 1e12 / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                       "end_to_end_frac": flop * steps / (wall / 1.0) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                       "what": "2 n_from n_to d flops of exact fp32 products (v_mfma_f32_32x32x2_f32) over the GEMM panels' "
                               "summed launch time; end_to_end_frac = over the whole step (row top-n included)"}
    if not args.no_cpu_baseline:
        # CPU arm + parity on a bounded random sample: float64 BLAS cosine + canonical top-n (oracle/dense.py)
        import oracle
        rows = np.sort(rng.choice(n_from, min(n_from, 1024), replace=False))     # (round 5: 64)
        c0 = time.perf_counter()
        e_idx, e_val = oracle.dense_cossim_topn(a[rows], b, top_n, 0.0, chunk_rows=128)
        dt = time.perf_counter() - c0
        err = float(np.abs(val[rows] - e_val).max())
        bad = int((idx[rows] != e_idx).any(axis=1).sum())
        out["cpu_baseline"] = {"value": len(rows) * float(n_to) / dt, "unit": "pairs/s", "cores": int(os.cpu_count() or 1), "kind": "port",
                               "sample": f"{len(rows)} random from-rows x all {n_to} to-vectors, oracle/dense.py (float64 numpy / BLAS "
                                         f"incl. the float64 conversion of the to-side), {dt:.1f} s"}
        out["parity_check"] = {"rows_checked": int(len(rows)), "rows": "seeded random sample of the from-rows",
                               "rows_with_index_diff": bad, "max_abs_score_err": err,
                               "planted_match_found_top1": float((idx[:, 0] == pick).mean()), "ok": bool(err <= 1e-5 and bad == 0)}
    return out


RUNNERS = {"c2": run_c2, "editdistance": run_editdistance, "rapidfuzz": run_rapidfuzz, "dense": run_dense, "tfidf_1m": run_tfidf_1m}


def sub_records(world, ctx, args):
    """Every other BASELINE configuration as a compact sub-record of the default line (N = 1).  A configuration that
    fails reports its error instead of taking the headline down."""
    plan = (("c2_tfidf_10k", lambda: run_c2(world, ctx, args)),
            ("editdistance", lambda: run_editdistance(world, ctx, args, steps=20, warmup=3, cpu_seconds=min(args.cpu_seconds, 4.0))),
            ("rapidfuzz_wratio", lambda: run_rapidfuzz(world, ctx, args, cpu_seconds=min(args.cpu_seconds, 6.0))),
            ("dense_shard", lambda: run_dense(world, ctx, args, steps=3, warmup=1)),
            ("tfidf_1m_shard", lambda: run_tfidf_1m(world, ctx, args)))
    out = {}
    for name, fn in plan:
        t0 = time.perf_counter()
        try:
            out[name] = compact(fn(), extra=("match_wall_ms_to_list_resident", "host_generation_s"))
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        out[name]["bench_wall_s"] = round(time.perf_counter() - t0, 1)
    return out


def run_rank(world, ctx, args):
    if args.config == "tfidf":
        out = headline(world, ctx, args)
        if out is not None and world.size == 1 and not args.no_configs:
            out["configs"] = sub_records(world, ctx, args)
    else:
        out = RUNNERS[args.config](world, ctx, args)
    world.barrier(ctx)
    return out


def main():
    args = parse()
    import polyfuzz_amd
    from polyfuzz_amd import _lib
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.transport == "local" and args.gpus > 1:
        n_dev = max(1, _lib.device_count())
        ctxs = [polyfuzz_amd.Context(r % n_dev) for r in range(args.gpus)]
        shared = {"barrier": threading.Barrier(args.gpus), "vals": [0.0] * args.gpus, "comms": _lib.Comm.local_group(ctxs)}
        outs, errs = [None] * args.gpus, [None] * args.gpus

        def rank_main(r):
            try:
                outs[r] = run_rank(LocalWorld(r, args.gpus, shared), ctxs[r], args)
            except BaseException as e:      # a dead rank must not leave the others waiting at a barrier
                errs[r] = e
                shared["barrier"].abort()

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(args.gpus)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        first = next((e for e in errs if e is not None and not isinstance(e, threading.BrokenBarrierError)), None) or \
            next((e for e in errs if e is not None), None)
        if first is not None:
            raise first
        out = outs[0]
        out["config"]["devices_visible"] = n_dev
    else:
        if env_world != args.gpus:
            if "WORLD_SIZE" not in os.environ and args.gpus > 1:
                raise SystemExit(self_launch(args.gpus))
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={env_world}")
        if args.rehearse_cpu:
            # (tests only: the rank logic behind the launcher on a box without a GPU -- the stand-in for the device, and with it
            # every use of oracle/, lives under tests/: tests/bench_rehearsal.py)
            import importlib
            world = TorchWorld(need_devices=False)
            out = importlib.import_module("tests.bench_rehearsal").rehearse_cpu(world, args, timed_steps)
            world.close()
            if out is not None:
                print(json.dumps(out))
            return
        world = TorchWorld() if env_world > 1 else World()
        # one process, one GPU: the matchers' own default context, so that the library's kernel timers see their launches too
        ctx = polyfuzz_amd.Context(world.local_rank) if env_world > 1 else polyfuzz_amd.Context.default()
        out = run_rank(world, ctx, args)
        if env_world > 1:
            if out is not None:
                out["config"]["rccl"] = world.rccl
                out["config"]["rendezvous"] = "torch.distributed over gloo (host): id broadcast, barrier, max over ranks; RCCL is the library's alone"
            world.close()
    if out is not None:
        if args.small:
            out["config"]["rehearsal"] = "--small: reduced sizes, NOT the BASELINE configuration"
        print(json.dumps(out))


if __name__ == "__main__":
    main()
