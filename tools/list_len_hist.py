#!/usr/bin/env python3
"""Design aid: distribution of the (n-gram, to-block) posting-list lengths a from-row walks in K3."""
import numpy as np
import scipy.sparse as sp
z = np.load("/tmp/sim/mats.npz")
A = sp.csr_matrix((z["ad"], z["ai"], z["ap"]), shape=tuple(z["ashape"]))
B = sp.csr_matrix((z["bd"], z["bi"], z["bp"]), shape=tuple(z["bshape"]))
n_to, V = B.shape
for C in (2048, 4096):
    nb = (n_to + C - 1) // C
    Bc = B.tocsc()
    blk = Bc.indices // C
    # per (k, b) list length table
    tab = np.zeros((V, nb), np.int32)
    cols = np.repeat(np.arange(V), np.diff(Bc.indptr))
    np.add.at(tab, (cols, blk), 1)
    rng = np.random.default_rng(0)
    rows = rng.choice(A.shape[0], 3000, replace=False)
    edges = [0, 1, 3, 5, 9, 17, 33, 65, 129, 10 ** 9]
    hist = np.zeros(len(edges) - 1)
    post = np.zeros(len(edges) - 1)
    chunks = 0
    nblocks = 0
    for i in rows:
        ks = A.indices[A.indptr[i]:A.indptr[i + 1]]
        L = tab[ks]            # [nnz, nb]
        h, _ = np.histogram(L, bins=edges)
        hist += h
        for q in range(len(edges) - 1):
            m = (L >= edges[q]) & (L < edges[q + 1])
            post[q] += L[m].sum()
        chunks += ((L + 63) // 64).sum()
        nblocks += nb
    print(f"C={C}: per row-block: lists by length")
    names = ["0", "1-2", "3-4", "5-8", "9-16", "17-32", "33-64", "65-128", ">128"]
    for n, h, p in zip(names, hist, post):
        print(f"   len {n:7s}: {h / nblocks:6.2f} lists  {p / nblocks:8.1f} postings")
    print(f"   chunks/block {chunks / nblocks:.2f}, postings/block {post.sum() / nblocks:.0f}")
