#!/usr/bin/env python3
"""CPU simulation of per-row adaptive MaxScore pruning for K3 (design aid, no GPU).

Per from-row the n-grams are ordered by to-side posting count (desc).  In every to-block the longest
prefix whose upper bound  sum a_k * max_b(k)  stays <= alpha * theta  (theta = running n-th best exact
score) is deferred: its lists are not scattered; to-rows whose partial sum reaches theta - UB are
completed by look-ups.  Reports the fraction of postings skipped and the look-ups added.
"""
import argparse
import os
import sys

import numpy as np
import scipy.sparse as sp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sample", type=int, default=400)
    ap.add_argument("--block", type=int, default=2048)
    ap.add_argument("--top-n", type=int, default=5)
    ap.add_argument("--alpha", type=float, nargs="+", default=[0.5, 0.75, 1.0])
    ap.add_argument("--heavy", type=int, default=0, help="only the H n-grams with the longest posting lists may be deferred (0 = any)")
    ap.add_argument("--cs-only", action="store_true", help="bound = ||a_N|| only (no per-n-gram max table)")
    ap.add_argument("--cache", default="/tmp/sim/mats.npz")
    args = ap.parse_args()
    z = np.load(args.cache)
    A = sp.csr_matrix((z["ad"], z["ai"], z["ap"]), shape=tuple(z["ashape"]))
    B = sp.csr_matrix((z["bd"], z["bi"], z["bp"]), shape=tuple(z["bshape"]))
    n_to, V = B.shape
    C = args.block
    nb = (n_to + C - 1) // C
    Bc = B.tocsc()
    df_to = np.diff(Bc.indptr)
    gmax = np.zeros(V)
    np.maximum.at(gmax, B.indices, B.data)
    blk_of = np.arange(n_to) // C
    is_heavy = np.ones(V, bool)
    if args.heavy:
        is_heavy[:] = False
        is_heavy[np.argsort(-df_to, kind='stable')[:args.heavy]] = True
    rng = np.random.default_rng(0)
    rows = rng.choice(A.shape[0], size=args.sample, replace=False)
    for alpha in args.alpha:
        tot_post = tot_skip = tot_cand = tot_look = 0
        for i in rows:
            ks = A.indices[A.indptr[i]:A.indptr[i + 1]]
            av = A.data[A.indptr[i]:A.indptr[i + 1]]
            if len(ks) == 0:
                continue
            order = np.argsort(-df_to[ks], kind="stable")
            ks, av = ks[order], av[order]
            # per-n-gram contribution vectors (dense over to-rows), cumulative from the light end
            contrib = np.zeros((len(ks), n_to))
            cnt = np.zeros((len(ks), nb), np.int64)
            for t, k in enumerate(ks):
                s, e = Bc.indptr[k], Bc.indptr[k + 1]
                contrib[t, Bc.indices[s:e]] = av[t] * Bc.data[s:e]
                cnt[t] = np.bincount(blk_of[Bc.indices[s:e]], minlength=nb)
            full = contrib.sum(axis=0)
            # suffix sums: light_from[t] = sum of contributions of n-grams t.. (those NOT deferred when prefix t is deferred)
            suffix = np.cumsum(contrib[::-1], axis=0)[::-1]
            ubs = np.sqrt(np.cumsum(av * av)) if args.cs_only else np.minimum(np.cumsum(av * gmax[ks]), np.sqrt(np.cumsum(av * av)))   # ub of deferring prefix [0..t]
            can = is_heavy[ks]
            best = np.zeros(0)
            theta = 0.0
            for b in range(nb):
                lo, hi = b * C, min(n_to, (b + 1) * C)
                tot_post += cnt[:, b].sum()
                # longest prefix with ub <= alpha*theta (and allowed)
                t = 0
                while t < len(ks) and can[t] and ubs[t] <= alpha * theta and theta > 0:
                    t += 1
                if t > 0:
                    ub = ubs[t - 1]
                    part = suffix[t, lo:hi] if t < len(ks) else np.zeros(hi - lo)
                    cands = np.count_nonzero((part > 0) & (part >= theta - ub))
                    tot_skip += cnt[:t, b].sum()
                    tot_cand += cands
                    tot_look += cands * t
                blk = full[lo:hi]
                best = np.sort(np.concatenate([best, blk[blk > 0]]))[::-1][:args.top_n]
                if len(best) == args.top_n:
                    theta = best[-1]
        print(f"alpha={alpha}: skipped {tot_skip / tot_post:.3f} of postings, candidates/row {tot_cand / len(rows):.1f}, "
              f"look-ups/row {tot_look / len(rows):.1f}, postings/row {tot_post / len(rows):.0f}")


if __name__ == "__main__":
    main()
