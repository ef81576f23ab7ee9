#!/usr/bin/env python3
"""CPU simulation (design aid): pruning with to-rows SORTED by the norm of their heavy part.

hn_j = || b_j restricted to the H heaviest n-grams ||.  With the to-rows ordered by hn (descending) every
to-block has a tight bound  ub_b = ||a_N|| * max_{j in block} hn_j  for the deferred set N.
"""
import argparse
import numpy as np
import scipy.sparse as sp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sample", type=int, default=300)
    ap.add_argument("--block", type=int, default=2048)
    ap.add_argument("--top-n", type=int, default=5)
    ap.add_argument("--heavy", type=int, default=64)
    ap.add_argument("--alpha", type=float, nargs="+", default=[0.5, 0.75, 1.0])
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--cache", default="/tmp/sim/mats.npz")
    args = ap.parse_args()
    z = np.load(args.cache)
    A = sp.csr_matrix((z["ad"], z["ai"], z["ap"]), shape=tuple(z["ashape"]))
    B = sp.csr_matrix((z["bd"], z["bi"], z["bp"]), shape=tuple(z["bshape"]))
    n_to, V = B.shape
    C = args.block
    nb = (n_to + C - 1) // C
    df_to = np.bincount(B.indices, minlength=V)
    order_h = np.argsort(-df_to, kind="stable")[:args.heavy]
    hid = -np.ones(V, np.int64)
    hid[order_h] = np.arange(args.heavy)
    heavy_col = hid >= 0
    BH = B.multiply(sp.csr_matrix(heavy_col.astype(np.float64))).tocsr()
    hn = np.sqrt(np.asarray(BH.multiply(BH).sum(axis=1)).ravel())
    perm = np.arange(n_to) if args.no_sort else np.argsort(-hn, kind="stable")
    B = B[perm]
    hn = hn[perm]
    Bc = B.tocsc()
    blk_of = np.arange(n_to) // C
    hnmax = np.array([hn[b * C:(b + 1) * C].max() for b in range(nb)])
    print("rows without heavy n-grams: %.3f; hnmax per block (first, median, last): %.3f %.3f %.3f"
          % ((hn == 0).mean(), hnmax[0], np.median(hnmax), hnmax[-1]))
    rng = np.random.default_rng(0)
    rows = rng.choice(A.shape[0], size=args.sample, replace=False)
    for alpha in args.alpha:
        tot_post = tot_skip = tot_cand = tot_look = 0
        steps_with_cand = steps = 0
        for i in rows:
            ks = A.indices[A.indptr[i]:A.indptr[i + 1]]
            av = A.data[A.indptr[i]:A.indptr[i + 1]]
            if len(ks) == 0:
                continue
            # heavy first (by slot), then the rest
            key = np.where(hid[ks] >= 0, hid[ks], 10 ** 6)
            order = np.argsort(key, kind="stable")
            ks, av = ks[order], av[order]
            nh = int((hid[ks] >= 0).sum())
            contrib = np.zeros((len(ks), n_to))
            cnt = np.zeros((len(ks), nb), np.int64)
            for t, k in enumerate(ks):
                s, e = Bc.indptr[k], Bc.indptr[k + 1]
                contrib[t, Bc.indices[s:e]] = av[t] * Bc.data[s:e]
                cnt[t] = np.bincount(blk_of[Bc.indices[s:e]], minlength=nb)
            full = contrib.sum(axis=0)
            suffix = np.cumsum(contrib[::-1], axis=0)[::-1]
            an = np.sqrt(np.cumsum(av * av))
            best = np.zeros(0)
            theta = 0.0
            for b in range(nb):
                lo, hi = b * C, min(n_to, (b + 1) * C)
                tot_post += cnt[:, b].sum()
                t = 0
                while t < nh and an[t] * hnmax[b] <= alpha * theta and theta > 0:
                    t += 1
                if t > 0:
                    ub = an[t - 1] * hnmax[b]
                    part = suffix[t, lo:hi] if t < len(ks) else np.zeros(hi - lo)
                    cm = (part > 0) & (part >= theta - ub)
                    cands = np.count_nonzero(cm)
                    tot_skip += cnt[:t, b].sum()
                    tot_cand += cands
                    tot_look += cands * t
                    # sweep steps (512 columns each) that see at least one candidate
                    for s0 in range(0, hi - lo, 512):
                        steps += 1
                        steps_with_cand += cm[s0:s0 + 512].any()
                blk = full[lo:hi]
                best = np.sort(np.concatenate([best, blk[blk > 0]]))[::-1][:args.top_n]
                if len(best) == args.top_n:
                    theta = best[-1]
        print(f"alpha={alpha}: skipped {tot_skip / tot_post:.3f} of postings, candidates/row {tot_cand / len(rows):.1f}, "
              f"look-ups/row {tot_look / len(rows):.1f}, sweep steps with a candidate {steps_with_cand / max(1, steps):.3f}")


if __name__ == "__main__":
    main()
