#!/usr/bin/env python3
"""CPU simulation of heavy-n-gram pruning for K3 (design aid, no GPU).

For a sample of from-rows of the bench workload, walks the to-blocks like the kernel does and counts the
postings a MaxScore-style rule would skip: the H n-grams with the longest posting lists are "heavy"; in a
block where the running n-th best exact score theta exceeds the row's upper bound of the heavy part
(sum a_k * max b_k), heavy lists are not scattered, and only to-rows whose light partial sum reaches
theta - UB are completed with look-ups.
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyfuzz_amd import synth                      # noqa: E402
from oracle.tfidf_oracle import TfidfOracle         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--sample", type=int, default=1500)
    ap.add_argument("--block", type=int, default=2048)
    ap.add_argument("--top-n", type=int, default=5)
    ap.add_argument("--heavy", type=int, nargs="+", default=[16, 32, 64])
    ap.add_argument("--cache", default="/tmp/sim/mats.npz")
    args = ap.parse_args()
    if os.path.exists(args.cache):
        z = np.load(args.cache)
        A = sp.csr_matrix((z["ad"], z["ai"], z["ap"]), shape=tuple(z["ashape"]))
        B = sp.csr_matrix((z["bd"], z["bi"], z["bp"]), shape=tuple(z["bshape"]))
    else:
        t0 = time.time()
        to_list = synth.company_names(args.n, seed=5678)
        from_list = synth.company_names(args.n, seed=1234)
        vec = TfidfOracle()
        vec.fit(to_list + from_list)
        V = len(vec.vocabulary)
        ip, ii, dd = vec.transform(to_list)
        B = sp.csr_matrix((dd, ii, ip), shape=(len(to_list), V))
        ip, ii, dd = vec.transform(from_list)
        A = sp.csr_matrix((dd, ii, ip), shape=(len(from_list), V))
        np.savez(args.cache, ad=A.data, ai=A.indices, ap=A.indptr, ashape=A.shape, bd=B.data, bi=B.indices,
                 bp=B.indptr, bshape=B.shape)
        print(f"vectorised in {time.time() - t0:.1f}s")
    n_to, V = B.shape
    C = args.block
    nb = (n_to + C - 1) // C
    df_to = np.bincount(B.indices, minlength=V)
    Bc = B.tocsc()
    rng = np.random.default_rng(0)
    rows = rng.choice(A.shape[0], size=args.sample, replace=False)
    # per (k, block) posting counts for the sampled rows are computed on the fly
    blk_of = np.arange(n_to) // C
    for H in args.heavy:
        heavy_ids = np.argsort(-df_to)[:H]
        is_heavy = np.zeros(V, bool)
        is_heavy[heavy_ids] = True
        # per heavy n-gram, per block max value
        hmax_blk = np.zeros((V, nb)) if False else None
        colmax_blk = {}
        for k in heavy_ids:
            s, e = Bc.indptr[k], Bc.indptr[k + 1]
            m = np.zeros(nb)
            np.maximum.at(m, blk_of[Bc.indices[s:e]], Bc.data[s:e])
            colmax_blk[k] = m
        BL = B.multiply(sp.csr_matrix((~is_heavy).astype(np.float64))).tocsr()
        BL.eliminate_zeros()
        tot_post = tot_skipped = tot_cand = tot_fix = 0
        tot_post_g = tot_skipped_g = tot_cand_g = 0
        full_rows = 0
        blocks_pruned = blocks_touched = 0
        for i in rows:
            ks = A.indices[A.indptr[i]:A.indptr[i + 1]]
            av = A.data[A.indptr[i]:A.indptr[i + 1]]
            if len(ks) == 0:
                continue
            full = np.asarray((B @ A[i].T).todense()).ravel()
            light = np.asarray((BL @ A[i].T).todense()).ravel()
            hk = [(k, a) for k, a in zip(ks, av) if is_heavy[k]]
            # postings per block: light and heavy
            cnt_l = np.zeros(nb, np.int64)
            cnt_h = np.zeros(nb, np.int64)
            for k in ks:
                s, e = Bc.indptr[k], Bc.indptr[k + 1]
                c = np.bincount(blk_of[Bc.indices[s:e]], minlength=nb)
                if is_heavy[k]:
                    cnt_h += c
                else:
                    cnt_l += c
            for variant in (0, 1):   # 0: per-block max bound, 1: global max bound
                best = np.zeros(0)
                theta = 0.0
                pruned_any = False
                for b in range(nb):
                    lo, hi = b * C, min(n_to, (b + 1) * C)
                    if variant == 0:
                        ub = sum(a * colmax_blk[k][b] for k, a in hk)
                    else:
                        ub = sum(a * colmax_blk[k].max() for k, a in hk)
                    post_b = cnt_l[b] + cnt_h[b]
                    prune = len(hk) > 0 and theta > ub and cnt_h[b] > 0
                    if variant == 0:
                        tot_post += post_b
                        blocks_touched += post_b > 0
                    else:
                        tot_post_g += post_b
                    if prune:
                        cands = np.count_nonzero((light[lo:hi] > 0) & (light[lo:hi] >= theta - ub))
                        if variant == 0:
                            tot_skipped += cnt_h[b]
                            tot_cand += cands
                            tot_fix += cands * len(hk)
                            blocks_pruned += 1
                        else:
                            tot_skipped_g += cnt_h[b]
                            tot_cand_g += cands
                        pruned_any = True
                    blk = full[lo:hi]
                    best = np.sort(np.concatenate([best, blk[blk > 0]]))[::-1][:args.top_n]
                    if len(best) == args.top_n:
                        theta = best[-1]
                if variant == 0 and not pruned_any:
                    full_rows += 1
        print(f"H={H:3d}: per-block bound: skipped {tot_skipped / tot_post:.3f} of postings, "
              f"candidates/row {tot_cand / len(rows):.1f}, look-ups/row {tot_fix / len(rows):.1f}, "
              f"blocks pruned {blocks_pruned / max(1, blocks_touched):.3f}, rows never pruned {full_rows / len(rows):.3f} | "
              f"global bound: skipped {tot_skipped_g / tot_post_g:.3f}, candidates/row {tot_cand_g / len(rows):.1f}")
        print(f"       postings/row {tot_post / len(rows):.0f}")


if __name__ == "__main__":
    main()
