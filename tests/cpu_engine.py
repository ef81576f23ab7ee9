"""Test double of polyfuzz_amd.pipeline.HipEngine for CPU-only boxes (TEST INFRASTRUCTURE).

Same operations, done by the oracle (float64) on host lists, with the cross-rank exchanges of the sharded
fit / the result all-gather over torch.distributed (gloo).  It exists so that the shard logic of
`TfidfMatchJob` -- the code of polyfuzz_amd/pipeline.py itself -- runs at world_size 2 where there is no GPU.
The exchanges mirror pfz_tfidf_fit_sharded (csrc/k1_vectorize.hip fit_impl): union of the ranks'
vocabularies (device: all-gather of code bitmaps + OR), sum of df / n_docs with the replicated list counted
by rank 0 only (device: all-reduce), equal-sized all-gather of the padded top-n blocks.
"""
import numpy as np

import oracle


class GlooComm:
    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()


class _Vec:
    def __init__(self, v):
        self.v = v

    def info(self):
        return {"vocab": len(self.v.vocabulary), "n_docs": self.v.n_docs, "code_bits": 0}


class _Csr:
    def __init__(self, triple, n_cols):
        self.triple, self.n_cols = triple, n_cols

    def download(self):
        p, i, v = self.triple
        return p, i, v, self.n_cols


class _TopN:
    def __init__(self, n_rows, ntop):
        self.n_rows, self.ntop = n_rows, ntop
        self.idx = np.full((n_rows, ntop), -7, np.int32)     # garbage until cleared / written
        self.val = np.full((n_rows, ntop), -7.0, np.float64)

    def clear(self):
        self.idx[:] = -1
        self.val[:] = 0.0

    def download(self):
        return self.idx.copy(), self.val.copy()


class OracleEngine:
    def __init__(self, torch=None):
        self.torch = torch

    def _oracle(self, params):
        return oracle.TfidfOracle(n_gram_range=(params.ngram_lo, params.ngram_hi), clean=bool(params.clean),
                                  remove_space_ngrams=bool(params.remove_space_ngrams))

    def upload_strings(self, strings):
        return list(strings)

    def fit(self, params, docs_a, docs_b):
        docs = list(docs_a or []) + (list(docs_b) if docs_b is not None and docs_b is not docs_a else [])
        return _Vec(self._oracle(params).fit(docs))

    def fit_sharded(self, comm, params, replicated, local_shard):
        dist, torch = comm.dist, self.torch
        v = self._oracle(params)
        counted = (list(replicated) if comm.rank == 0 else []) + list(local_shard)
        local_df = {}
        for s in counted:
            for g in set(v._analyze(s)):
                local_df[g] = local_df.get(g, 0) + 1
        local_vocab = set(local_df) | {g for s in replicated for g in v._analyze(s)}
        gathered = [None] * comm.world
        dist.all_gather_object(gathered, sorted(local_vocab))
        vocab = sorted(set().union(*gathered))
        df = torch.tensor([local_df.get(g, 0) for g in vocab], dtype=torch.int64)
        n_docs = torch.tensor([len(counted)], dtype=torch.int64)
        dist.all_reduce(df)
        dist.all_reduce(n_docs)
        v.vocabulary = vocab
        v.index = {g: i for i, g in enumerate(vocab)}
        v.df = df.numpy()
        v.n_docs = int(n_docs.item())
        v.idf = np.log((v.n_docs + 1.0) / (v.df.astype(np.float64) + 1.0)) + 1.0
        return _Vec(v)

    def transform(self, vec, docs):
        return _Csr(vec.v.transform(docs), len(vec.v.vocabulary))

    def build_index(self, to_csr):
        return to_csr

    def alloc_topn(self, n_rows, ntop):
        return _TopN(n_rows, ntop)

    def cossim_topn(self, index, from_csr, ntop, lower_bound, exclude_diag, diag_offset, out):
        n = len(from_csr.triple[0]) - 1
        idx, val = oracle.cossim_topn(from_csr.triple, index.triple, index.n_cols, ntop, lower_bound)
        if exclude_diag:        # the oracle's exclude_diag drops j == i; a shard needs j == i + diag_offset
            idx, val = oracle.cossim_topn(from_csr.triple, index.triple, index.n_cols, ntop + 1, lower_bound)
            keep_i = np.full((n, ntop), -1, np.int32)
            keep_v = np.zeros((n, ntop))
            for i in range(n):
                sel = [(j, s) for j, s in zip(idx[i], val[i]) if j != i + diag_offset and j >= 0][:ntop]
                for r, (j, s) in enumerate(sel):
                    keep_i[i, r], keep_v[i, r] = j, s
            idx, val = keep_i, keep_v
        out.idx[:n], out.val[:n] = idx, val      # rows beyond the shard (padding) are not touched
        return out

    def allgather_topn(self, comm, local, out):
        torch = self.torch
        li, lv = torch.from_numpy(local.idx), torch.from_numpy(local.val)
        gi = [torch.empty_like(li) for _ in range(comm.world)]
        gv = [torch.empty_like(lv) for _ in range(comm.world)]
        comm.dist.all_gather(gi, li)
        comm.dist.all_gather(gv, lv)
        out.idx[:] = torch.cat(gi).numpy()
        out.val[:] = torch.cat(gv).numpy()
        return out

    # ---- the self-match cut over the ranks in K3's symmetric form (polyfuzz_amd/csrc/k3_symmetric.hip, k3_sym_sharded) ----
    SYM_BLOCK = 16         # to-rows per block of the emulation (the kernel: 2048)

    force_row_shards_on_rank = None        # tests: this rank alone says no (different environment / failed allocation)

    def symmetric_ok(self, comm, index, csr, ntop):
        """the collective question (pfz_comm_symmetric_ok): every rank's own answer, AND-ed over the ranks"""
        mine = len(csr.triple[0]) - 1 > 2 * self.SYM_BLOCK and comm.rank != self.force_row_shards_on_rank
        torch = self.torch
        t = torch.tensor([int(mine)], dtype=torch.int32)
        comm.dist.all_reduce(t, op=comm.dist.ReduceOp.MIN)
        return bool(t.item())

    def cossim_topn_symmetric(self, comm, index, csr, ntop, lower_bound, out):
        """The partition rule of the device job, on oracle scores: rank r works on the rows r, r + world, ...; a pair inside one
        block is scored by both its rows (each keeps it), a pair of different blocks ONCE, by the row of the lower block, which
        keeps it and hands it to the other row; every rank cuts what it knows of a row to ntop, the lists are all-gathered and
        merged by (score desc, index asc)."""
        n = len(csr.triple[0]) - 1
        c = self.SYM_BLOCK
        idx, val = oracle.cossim_topn(csr.triple, csr.triple, csr.n_cols, max(n - 1, 1), lower_bound, exclude_diag=True)
        lists = [[] for _ in range(n)]
        for j in range(comm.rank, n, comm.world):
            for i, s in zip(idx[j], val[j]):
                if i < 0:
                    continue
                if i // c == j // c:
                    lists[j].append((-s, int(i)))
                elif i // c > j // c:
                    lists[j].append((-s, int(i)))
                    lists[int(i)].append((-s, j))
        mine = [sorted(lst)[:ntop] for lst in lists]
        gathered = [None] * comm.world
        comm.dist.all_gather_object(gathered, mine)
        out.clear()
        for i in range(n):
            best = sorted(set(e for part in gathered for e in part[i]))[:ntop]
            for r, (ns, j) in enumerate(best):
                out.idx[i, r], out.val[i, r] = j, -ns
        return out
