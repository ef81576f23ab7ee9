"""Device-resident TF-IDF match job: the hot path of `TFIDF.match` with the string
lists already in HBM and the result left in HBM.

One `step()` = everything the reference does between receiving the lists and
building the DataFrame (reference polyfuzz/models/_tfidf.py:93-98 ->
_utils.py:54-102): fit the vocabulary/idf on to + from, vectorise both lists,
build the to-side inverted index, run the fused cosine top-n.

Multi-GPU (one process per GPU, or one process driving several contexts): the
from-list is row-sharded (each rank holds its shard), the to-list is replicated.
The fit is exact: vocabulary bitmaps are all-gathered and document frequencies
all-reduced, so every rank ends with the vectoriser a single-GPU fit on the
concatenated lists would produce; the per-shard top-n blocks -- padded to the
largest shard so that the all-gather moves equal blocks -- are all-gathered into
the full result on every rank.

The job talks to the device through an *engine* object (`HipEngine`, the C ABI
of include/polyfuzz_hip.h).  The seam exists so that the shard logic of this
file -- which rank fits on what, padding, diagonal offsets, the order of the
exchanges -- can be driven at world_size 2 on a CPU-only box by a test double
(tests/cpu_engine.py: oracle arithmetic + gloo); the product has one engine.
"""
import numpy as np

from . import _lib

PROFILED_KERNELS = ("k1_extract", "k2_rows_short", "k2_rows_long", "k2_df_hist", "k2_finalize", "k_index_count",
                    "k_index_fill", "k_index_bank_order", "k3_cossim_topn")


def shard_bounds(n, world, rank):
    """Contiguous row shard [begin, end) of rank: sizes differ by at most one."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def balanced_bounds(strings, world):
    """Contiguous row shards [(begin, end)] * world of a from-list, cut so that every shard holds the same number of
    CHARACTERS rather than the same number of rows.  A from-row's cost in K3 is a constant (the accumulator sweeps) plus
    the postings of its n-grams, in K4 / K7 its length times the to-list's characters -- both grow with the string's
    length, and a sorted list is skewed (the reference's company names: equal row counts leave the slowest of 8 shards
    11 % above the mean by the LDS-floor model, equal characters 2 %; tools/predict_scaling.py measures it).  Every rank
    computes the same cuts from the same list; shards stay contiguous, so results concatenate in order."""
    n = len(strings)
    if world <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (max(world, 1) - 1)
    cost = np.fromiter((len(s) + 1 for s in strings), np.int64, n)
    cs = np.cumsum(cost)
    edges = [0] + [int(np.searchsorted(cs, cs[-1] * r // world, side="left")) for r in range(1, world)] + [n]
    edges = [min(max(e, 0), n) for e in edges]
    for i in range(1, len(edges)):           # monotone (degenerate lists)
        edges[i] = max(edges[i], edges[i - 1])
    return list(zip(edges[:-1], edges[1:]))


class HipEngine:
    """The device operations a match job is made of, on libpolyfuzz_hip.so."""

    def __init__(self, ctx):
        self.ctx = ctx

    def upload_strings(self, strings):
        return _lib.DeviceStrings.upload(self.ctx, strings)

    def fit(self, params, docs_a, docs_b):
        return _lib.DeviceTfidf.fit(self.ctx, params, docs_a, docs_b)

    def fit_sharded(self, comm, params, replicated, local_shard):
        return _lib.tfidf_fit_sharded(self.ctx, comm, params, replicated, local_shard)

    def transform(self, vec, docs):
        return vec.transform(docs)

    def build_index(self, to_csr):
        return _lib.DeviceIndex.build(self.ctx, to_csr)

    def alloc_topn(self, n_rows, ntop):
        return _lib.DeviceTopN.alloc(self.ctx, n_rows, ntop)

    def cossim_topn(self, index, from_csr, ntop, lower_bound, exclude_diag, diag_offset, out):
        return _lib.cossim_topn(self.ctx, index, from_csr, ntop, lower_bound, exclude_diag=exclude_diag,
                                diag_offset=diag_offset, out=out)

    def allgather_topn(self, comm, local, out):
        return comm.allgather_topn(local, out)

    def symmetric_ok(self, comm, index, csr, ntop):
        """COLLECTIVE: every rank's own rule (sizes, environment) and allocations, AND-ed over the ranks (pfz_comm_symmetric_ok) --
        a rank alone must not decide which collectives the job issues"""
        return comm.symmetric_ok(index, csr, ntop)

    def cossim_topn_symmetric(self, comm, index, csr, ntop, lower_bound, out):
        return comm.cossim_topn_symmetric(index, csr, ntop, lower_bound, out)


class TfidfMatchJob:
    def __init__(self, ctx, from_shard, to_list, top_n=1, min_similarity=0.0, n_gram_range=(3, 3),
                 clean_string=True, remove_space_ngrams=True, comm=None, self_match=False, shard_offset=0,
                 rows_per_rank=None, engine=None):
        """self_match: the from-rows are rows [shard_offset, shard_offset + len(from_shard)) of to_list (the whole
        list, replicated); to_list=None = the from-list is the whole list (single GPU).
        rows_per_rank: size of the largest shard when the ranks' shards differ (shard_bounds); the
        per-rank result block is padded to it so that the all-gather moves equal blocks.
        comm: an object with .rank / .world (polyfuzz_amd.Comm)."""
        self.ctx = ctx
        self.eng = engine if engine is not None else HipEngine(ctx)
        self.comm = comm
        self.top_n = int(top_n)
        self.min_similarity = float(min_similarity)
        self.self_match = bool(self_match)
        self.shard_offset = int(shard_offset)
        self.params = _lib.TfidfParams(int(n_gram_range[0]), int(n_gram_range[1]), int(bool(clean_string)),
                                       int(bool(remove_space_ngrams)))
        self.n_from = len(from_shard)
        self.from_dev = self.eng.upload_strings(from_shard)
        if to_list is None:                    # single-list self-match: the list is its own to-side
            if not self.self_match or self.shard_offset != 0:
                raise ValueError("to_list=None means a whole-list self-match (self_match=True, shard_offset=0)")
            self.n_to, self.to_dev = self.n_from, self.from_dev
        elif to_list is from_shard and self.self_match and self.shard_offset == 0:
            # the rank's from-rows ARE the whole replicated list (bench.py's weak scaling: every rank its own copy of the
            # query batch): one upload, one vectorisation
            self.n_to, self.to_dev = self.n_from, self.from_dev
        else:
            self.n_to = len(to_list)
            self.to_dev = self.eng.upload_strings(to_list)
        self.rows_per_rank = self.n_from if rows_per_rank is None else int(rows_per_rank)
        if self.rows_per_rank < self.n_from:
            raise ValueError("rows_per_rank is smaller than this rank's shard")
        self.local = self.eng.alloc_topn(self.rows_per_rank, self.top_n)
        self.local.clear()                     # padding rows stay "no match"
        self.gathered = None
        if comm is not None and comm.world > 1:
            self.gathered = self.eng.alloc_topn(self.rows_per_rank * comm.world, self.top_n)
        self.vec = self.from_csr = self.to_csr = self.index = None
        # a self-match cut over several GPUs in K3's symmetric form leaves the FULL result (n_to x top_n) on every rank
        self.full = None
        self.result_is_full = False
        self._symmetric = None                 # the ranks' common answer to "symmetric form?", asked by the first step

    def step(self):
        eng = self.eng
        sharded = self.comm is not None and self.comm.world > 1
        if self.self_match:
            # reference _tfidf.py:113-116: a self-match fits on the list ALONE (n_docs = len(list)); the
            # replicated list is the whole list on every rank, so the fit needs no exchange
            self.vec = eng.fit(self.params, self.to_dev, None)
        elif sharded:
            self.vec = eng.fit_sharded(self.comm, self.params, self.to_dev, self.from_dev)
        else:
            self.vec = eng.fit(self.params, self.to_dev, self.from_dev)
        self.to_csr = eng.transform(self.vec, self.to_dev)
        self.index = eng.build_index(self.to_csr)
        if self.self_match and sharded and self.to_dev is not self.from_dev and self._symmetric is None:
            # asked once per job (its sizes are fixed), by all ranks together: ADVICE r5 -- ranks that read different environments
            # or whose allocations fail differently would otherwise issue different collectives and wait for each other for ever
            self._symmetric = bool(eng.symmetric_ok(self.comm, self.index, self.to_csr, self.top_n))
        if self._symmetric:
            # One list against itself, cut over the ranks (bench --scaling strong): the symmetric form of K3 scores every
            # unordered pair of rows ONCE over all ranks -- rank r works on the rows r, r + world, ... of the replicated list,
            # whatever contiguous shard it was handed (the sorted list's cost is spread evenly that way) -- and the ranks'
            # per-row candidate lists are all-gathered and merged on every rank: the full result, no second exchange.
            if self.full is None:
                self.full = eng.alloc_topn(self.n_to, self.top_n)
            self.from_csr = self.to_csr
            eng.cossim_topn_symmetric(self.comm, self.index, self.to_csr, self.top_n, self.min_similarity, self.full)
            self.result_is_full = True
            return self.full
        self.result_is_full = False
        if self.self_match and self.shard_offset == 0 and (self.to_dev is self.from_dev or (not sharded and self.n_from == self.n_to)):
            self.from_csr = self.to_csr            # the same rows: vectorise once (reference _tfidf.py:114-116)
        else:
            self.from_csr = eng.transform(self.vec, self.from_dev)
        eng.cossim_topn(self.index, self.from_csr, self.top_n, self.min_similarity, self.self_match, self.shard_offset,
                        self.local)
        if self.gathered is not None:
            eng.allgather_topn(self.comm, self.local, self.gathered)
        return self.gathered if self.gathered is not None else self.local

    def whole_result(self, idx, val, shard_sizes):
        """(idx, val) of ALL from-rows out of what the last step() returned and .download() gave"""
        if self.result_is_full:
            return idx, val
        return self.unpad(idx, val, shard_sizes, self.rows_per_rank)

    @staticmethod
    def unpad(idx, val, shard_sizes, rows_per_rank):
        """Drop the padding rows of an all-gathered result: (idx, val) of shape
        [world * rows_per_rank, top_n] -> [sum(shard_sizes), top_n]."""
        keep = np.concatenate([np.arange(r * rows_per_rank, r * rows_per_rank + n) for r, n in enumerate(shard_sizes)])
        return idx[keep], val[keep]

    def step_description(self):
        what = ("fit vocabulary+idf on the list (K1/K2), vectorise it, build the inverted index, fused cosine top-n "
                "with the diagonal excluded (K3)" if self.self_match else
                "fit vocabulary+idf on to+from (K1/K2), vectorise both lists, build the to-side inverted index, "
                "fused cosine top-n (K3)")
        if self.result_is_full:
            return what + ("; the self-match cut over the ranks in K3's symmetric form (every unordered pair once over all ranks): all-gather "
                           "of the ranks' pass-0 thresholds and of their per-row candidate lists, merged on every rank")
        return what + ("; all-gather of the per-shard results" if self.gathered else "")

    # ---- host-side accounting (never inside the timed region) ---------------------
    def host_matrices(self):
        """(from CSR triple, to CSR triple, n_cols) of the last step as float64 host arrays."""
        fp, fi, fv, n_cols = self.from_csr.download()
        tp, ti, tv, _ = self.to_csr.download()
        return (fp, fi, fv.astype(np.float64)), (tp, ti, tv.astype(np.float64)), n_cols

    def stats(self):
        (fp, fi, _), (tp, ti, _), n_cols = self.host_matrices()
        df_from = np.bincount(fi, minlength=n_cols).astype(np.float64)
        df_to = np.bincount(ti, minlength=n_cols).astype(np.float64)
        out = {"vocab": int(n_cols), "nnz_from": int(fp[-1]), "nnz_to": int(tp[-1]),
               "madds": float((df_from * df_to).sum())}
        if self.from_csr is self.to_csr and self.index is not None:
            # what the symmetric form of K3 (k3_symmetric.hip) executes for this self-match: the pairs inside a to-block in both
            # directions (pass 0), the pairs of different blocks once (pass 1); every row sweeps its own block and those above
            info = self.index.info()
            c, nb = info["block_cols"], info["n_blocks"]
            row_of = np.repeat(np.arange(len(tp) - 1, dtype=np.int64), np.diff(tp))
            per_list = np.bincount(ti.astype(np.int64) * nb + row_of // c, minlength=n_cols * nb).astype(np.float64)
            madds_diag = float((per_list * per_list).sum())
            out["madds_symmetric"] = 0.5 * (out["madds"] + madds_diag)
            out["cells_symmetric"] = float(((nb - np.arange(len(tp) - 1, dtype=np.int64) // c) * c).sum())
        return out


def sharded_self_match(ctx, comm, names, top_n=1, min_similarity=0.0, n_gram_range=(3, 3), clean_string=True,
                       remove_space_ngrams=True, engine=None):
    """`TFIDF(min_similarity, top_n, ...).match(names)` (reference _tfidf.py:68-100, a list against itself) on the GPUs of a
    communicator: EVERY rank calls it with the same list and gets the same, full frame -- Python list in, DataFrame out, the unit
    SURVEY section 8d's metric is defined on.  Every rank packs and uploads the list (it is replicated), works on its share of
    the rows (`TfidfMatchJob`: K3's symmetric form cut over the ranks where it applies, cost-balanced row shards otherwise), the
    exchange leaves the full result on every rank, and every rank builds the frame from it."""
    from .models._utils import topn_to_frame
    names = names if isinstance(names, list) else list(names)       # (no copy of a list: 100 000 references taken and dropped per call)
    # (row shards of equal ROWS up to the size K3's symmetric form takes -- it deals the rows r, r + world, ... itself and the cuts only
    # name what each rank uploads --, of equal characters beyond: balanced_bounds walks every string, 10 ms per 100 000 names, which
    # is three of these calls)
    bounds = balanced_bounds(names, comm.world) if len(names) > 250_000 else [shard_bounds(len(names), comm.world, r) for r in range(comm.world)]
    b, e = bounds[comm.rank]
    sizes = [y - x for x, y in bounds]
    job = TfidfMatchJob(ctx, names[b:e], names, top_n=top_n, min_similarity=min_similarity, n_gram_range=n_gram_range,
                        clean_string=clean_string, remove_space_ngrams=remove_space_ngrams, comm=comm, self_match=True,
                        shard_offset=b, rows_per_rank=max(sizes), engine=engine)
    idx, val = job.step().download()
    idx, val = job.whole_result(idx, val, sizes)
    return topn_to_frame(np.ascontiguousarray(idx, np.int32), np.ascontiguousarray(val, np.float32), names, names, top_n)


class ToShardedMatchJob:
    """The other sharding of the TF-IDF match: the TO-list is row-sharded over the ranks, the from-list replicated
    (BASELINE north_star's variant; needed when the to-side index does not fit one GPU).  Every rank fits the exact
    global vectoriser (sharded fit: the from-list counted once, its to-shard locally), indexes its to-shard, matches ALL
    from-rows against it, and the per-shard candidates are all-gathered and merged by (score desc, global index asc)."""

    def __init__(self, ctx, from_list, to_shard, to_offset, comm, top_n=1, min_similarity=0.0, n_gram_range=(3, 3),
                 clean_string=True, remove_space_ngrams=True, self_match=False):
        """self_match: from_list is the whole list and to_shard its rows [to_offset, to_offset + len(to_shard))."""
        self.ctx, self.comm = ctx, comm
        self.top_n, self.min_similarity, self.self_match = int(top_n), float(min_similarity), bool(self_match)
        self.to_offset = int(to_offset)
        self.params = _lib.TfidfParams(int(n_gram_range[0]), int(n_gram_range[1]), int(bool(clean_string)),
                                       int(bool(remove_space_ngrams)))
        self.from_dev = _lib.DeviceStrings.upload(ctx, from_list)
        self.to_dev = _lib.DeviceStrings.upload(ctx, to_shard)
        self.n_from = len(from_list)
        self.local = _lib.DeviceTopN.alloc(ctx, self.n_from, self.top_n)
        self.merged = _lib.DeviceTopN.alloc(ctx, self.n_from, self.top_n)

    def step(self):
        ctx = self.ctx
        # a self-match fits on the list alone (reference _tfidf.py:113-116): the shards ARE the list
        self.vec = _lib.tfidf_fit_sharded(ctx, self.comm, self.params, None if self.self_match else self.from_dev, self.to_dev)
        self.to_csr = self.vec.transform(self.to_dev)
        self.index = _lib.DeviceIndex.build(ctx, self.to_csr)
        self.from_csr = self.vec.transform(self.from_dev)
        _lib.cossim_topn(ctx, self.index, self.from_csr, self.top_n, self.min_similarity, exclude_diag=self.self_match,
                         diag_offset=-self.to_offset, out=self.local)
        return self.comm.merge_to_shards(self.local, self.to_offset, self.merged)


class DenseMatchJob:
    """Device-resident dense cosine top-n (K5) of a row shard of from-vectors against replicated to-vectors
    (reference _embeddings.py:127-133 -> _utils.py:74-77,94-102 on ready-made embeddings; BASELINE config 5:
    500k x 500k x 768 on 8 GPUs = 62.5k from-rows per rank).  Shards are independent; the only exchange is the
    all-gather of the padded per-shard top-n blocks."""

    def __init__(self, ctx, from_shard, to_vectors, top_n=1, min_similarity=0.0, normalize=True, comm=None,
                 self_match=False, shard_offset=0, rows_per_rank=None):
        self.ctx, self.comm = ctx, comm
        self.top_n, self.min_similarity = int(top_n), float(min_similarity)
        self.self_match, self.shard_offset = bool(self_match), int(shard_offset)
        self.from_dev = _lib.DeviceDense.upload(ctx, from_shard, normalize)
        self.to_dev = self.from_dev if to_vectors is None else _lib.DeviceDense.upload(ctx, to_vectors, normalize)
        if to_vectors is None and (not self.self_match or self.shard_offset != 0):
            raise ValueError("to_vectors=None means a whole-matrix self-match (self_match=True, shard_offset=0)")
        self.n_from, self.n_to = self.from_dev.n, self.to_dev.n
        self.rows_per_rank = self.n_from if rows_per_rank is None else int(rows_per_rank)
        if self.rows_per_rank < self.n_from:
            raise ValueError("rows_per_rank is smaller than this rank's shard")
        self.local = _lib.DeviceTopN.alloc(ctx, self.rows_per_rank, self.top_n)
        self.local.clear()
        self.gathered = None
        if comm is not None and comm.world > 1:
            self.gathered = _lib.DeviceTopN.alloc(ctx, self.rows_per_rank * comm.world, self.top_n)

    def step(self):
        _lib.dense_topn(self.ctx, self.from_dev, self.to_dev, self.top_n, self.min_similarity,
                        exclude_diag=self.self_match, diag_offset=self.shard_offset, out=self.local)
        if self.gathered is not None:
            self.comm.allgather_topn(self.local, self.gathered)
        return self.gathered if self.gathered is not None else self.local


class BestChoiceJob:
    """The edit-distance matchers on a row shard ("shard from-strings, replicate to-strings"):
    every rank scores its from-strings against the whole to-list -- K4 for ratio / QRatio / token_sort_ratio, K7 for
    WRatio and the other per-pair scorers (reference _distance.py:89-102, _rapidfuzz.py:99-113) -- and the per-shard
    (first best index, float64 score) blocks are all-gathered, padded to the largest shard.  The float64 scores travel
    as two 32-bit words in the value lanes of a two-column result buffer (the gather moves bytes), so every rank ends with
    the exact scores of all from-strings -- which is also all the reference's global min-max normalisation needs.
    Both lists are uploaded once, at construction; the to-side plan (K4) / the token forms and the plan (K7) are built on
    the device on the first step and stay cached on the handles: a step is device work only, its result stays on the
    device (`result_host` downloads it)."""

    def __init__(self, ctx, from_shard, to_list, scorer="ratio", comm=None, skip=None, rows_per_rank=None):
        from .models._rapidfuzz import _DEVICE_SCORERS, _K4_SCORERS, upload_for
        if scorer not in _DEVICE_SCORERS:
            raise NotImplementedError(f"scorer {scorer!r} has no kernel")
        self.ctx, self.comm, self.scorer = ctx, comm, scorer
        self.from_shard, self.to_list, self.skip = list(from_shard), to_list, skip
        self.n_from = len(self.from_shard)
        self.rows_per_rank = self.n_from if rows_per_rank is None else int(rows_per_rank)
        if self.rows_per_rank < self.n_from:
            raise ValueError("rows_per_rank is smaller than this rank's shard")
        self.k4 = scorer in _K4_SCORERS
        self._qfix = None
        self.f_dev = upload_for(ctx, scorer, self.from_shard)
        self.t_dev = upload_for(ctx, scorer, to_list)
        self.local = _lib.DeviceTopN.alloc(ctx, max(self.rows_per_rank, 1), 2)
        self.local.clear()                     # padding rows stay "no choice" (index -1, score 0.0)
        self.gathered = None
        if comm is not None and comm.world > 1:
            self.gathered = _lib.DeviceTopN.alloc(ctx, max(self.rows_per_rank, 1) * comm.world, 2)

    def plan_info(self):
        """the cached to-side plan: K4's (alphabet, groups, character steps) or K7's (alphabet, groups, tokens)"""
        return _lib.indel_plan_info(self.ctx, self.t_dev) if self.k4 else _lib.fuzz_plan_info(self.ctx, self.t_dev)

    def _qratio_rows(self):
        """QRatio = ratio except that an EMPTY from-string scores 0 against every choice (also the empty one, which ratio
        scores 100): its first best is simply its first choice.  [(local row, first choice)] of this shard's empty strings."""
        if self._qfix is None:
            self._qfix = []
            if self.scorer == "QRatio":
                for i, s in enumerate(self.from_shard):
                    if len(s) == 0:
                        first = next((j for j in range(len(self.to_list))
                                      if not (self.skip is not None and (j == self.skip[i] or j <= -2 - int(self.skip[i])))), -1)
                        self._qfix.append((i, first))
        return self._qfix

    def step(self):
        if self.n_from and len(self.to_list):
            if self.k4:
                _lib.indel_argmax_dev(self.ctx, self.f_dev, self.t_dev, self.local, self.skip)
            else:
                _lib.fuzz_extract_one_dev(self.ctx, self.f_dev, self.t_dev, self.scorer, self.local, self.skip)
            fix = self._qratio_rows()
            if fix:
                # patched in the shard's OWN block before the all-gather, so every rank ends with the same rows;
                # empty from-strings under QRatio are rare enough for a host round trip of the block
                idx, val = self.local.download()
                for i, first in fix:
                    idx[i, 0], val[i, :] = first, 0.0
                self.local.upload(idx, val)
        if self.gathered is not None:
            self.comm.allgather_topn(self.local, self.gathered)
        return self.gathered if self.gathered is not None else self.local

    def result_host(self, result):
        """(index int32[n], score float64[n]) of what step() returned (all ranks' rows, padded, when sharded)"""
        idx, score = _lib.best_from_topn(*result.download())
        if self.gathered is None:
            idx, score = idx[:self.n_from], score[:self.n_from]
        return idx, score

    def roofline(self, step_s, peak_tops):
        """K7's work accounting (one more step with the device counters on): every pair costs its upper bound -- 4 x 128-bit
        loads, 8 v_sad_u8, the float32 bound: priced at 40 integer operations -- and a scored pair its LCS passes /
        window sweeps -- priced at 10 int32 operations per 64-bit word-step (the recurrence's 5 on both halves), from
        the kernel's own estimate by the pair's lengths."""
        if self.k4:
            return None
        # (this rank's shard only, no exchange: the other ranks are not here)
        w = _lib.fuzz_extract_one_dev(self.ctx, self.f_dev, self.t_dev, self.scorer, self.local, self.skip, counters=True) \
            if self.n_from and len(self.to_list) else None
        w = w or {"pairs_bounded": 0, "pairs_scored": 0, "word_steps_scored": 0}
        pairs = self.n_from * float(len(self.to_list))
        ops = 40.0 * w["pairs_bounded"] + 10.0 * w["word_steps_scored"]
        return {"kernel": "k7_fuzz", "bound": "int32 VALU issue (+ LDS look-ups)", "achieved": ops / step_s / 1e12 if step_s > 0 else None,
                "peak": peak_tops, "unit": "Tera int-op/s", "frac": ops / step_s / 1e12 / peak_tops if step_s > 0 else None, "traffic": None,
                "pairs_per_s_kernel": pairs / step_s if step_s > 0 else None, "pairs_bounded": w["pairs_bounded"],
                "pairs_scored": w["pairs_scored"], "scored_fraction": w["pairs_scored"] / max(pairs, 1.0),
                "word_steps_scored_estimate": w["word_steps_scored"],
                "what": "40 int-ops per bounded pair (two sweeps) + 10 per 64-bit word-step of the scored pairs, over the summed "
                        "k7 kernel time, against 256 CU x 4 SIMD x 32 lanes x 2.4 GHz; extractOne keeps only the maximum, so the "
                        "kernel scores only the pairs whose upper bound reaches the best score found so far"}

    @staticmethod
    def unpad(idx, score, sizes, rows_per_rank):
        keep = np.concatenate([np.arange(r * rows_per_rank, r * rows_per_rank + m) for r, m in enumerate(sizes)])
        return idx[keep], score[keep]


def run_sharded_job(ctxs, comms, from_list, to_list, **job_kw):
    """One process, several contexts: run the row-sharded job with one host thread per rank and return
    (idx, val) of the full from-list (the ranks' all-gathered, un-padded result; identical on every rank).
    `comms` = polyfuzz_amd.Comm.local_group(ctxs).  self_match=True in job_kw: `to_list` is the whole list and
    `from_list` must be the same list (every rank matches its row shard of it)."""
    import concurrent.futures as cf
    world = len(ctxs)
    n = len(from_list)
    bounds = [shard_bounds(n, world, r) for r in range(world)]
    sizes = [e - b for b, e in bounds]
    rpr = max(sizes)
    self_match = bool(job_kw.get("self_match", False))

    def rank_fn(r):
        b, e = bounds[r]
        job = TfidfMatchJob(ctxs[r], from_list[b:e], to_list, comm=comms[r], rows_per_rank=rpr,
                            shard_offset=b if self_match else 0, **job_kw)
        out = job.step()
        idx, val = out.download()
        return job.whole_result(idx, val, sizes), job

    with cf.ThreadPoolExecutor(world) as ex:
        futs = [ex.submit(rank_fn, r) for r in range(world)]
        return [f.result(timeout=300) for f in futs]
