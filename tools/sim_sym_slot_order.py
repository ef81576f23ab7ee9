"""CPU simulation for K3's symmetric form (csrc/k3_symmetric.hip): what the second filter of pass 1 would see if the
to-rows of a block were re-dealt to the accumulator slots BY THRESHOLD after pass 0 -- one threshold per (sweep step,
lane) = the minimum over the eight slots the lane reads -- against what it sees today (one 16-bit threshold per row,
rows as they lie).  Counts, over a sample of (lower block, higher block) pairs of the 100 000 company names:
  * the share of (from-row, sweep step) with at least one hit (a "hit step" runs the rare path),
  * candidates handed over per from-row.
Orders compared: as they lie + exact per-row thresholds (today); as they lie + group minimum; full sort inside the
block; sort inside each bank class (slot mod 32 kept, so the index's bank order of the heavy lists survives).
Test infrastructure / design evidence, not product code.  Usage: python tools/sim_sym_slot_order.py [n_pairs]"""
import sys
import os
import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sklearn.feature_extraction.text import TfidfVectorizer
from oracle.tfidf_oracle import create_ngrams
from polyfuzz_amd.datasets import load_company_names

C = 2048


def lane_groups():
    """slot -> group id (step t, lane l): slots 512 t + 4 l + c and 512 t + 256 + 4 l + c, c = 0..3"""
    g = np.empty(C, np.int64)
    for s in range(C):
        t, rem = divmod(s, 512)
        half, q = divmod(rem, 256)
        g[s] = t * 64 + (q >> 2)
    return g


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    names = load_company_names()
    A = TfidfVectorizer(min_df=1, analyzer=create_ngrams).fit_transform(names).astype(np.float32).tocsr()
    n = A.shape[0]
    nb = (n + C - 1) // C
    ntop = 5
    # own-block thresholds (pass 0): the ntop-th best of the row inside its own block, the diagonal excluded
    thr = np.zeros(n, np.float32)
    for b in range(nb):
        lo, hi = b * C, min(n, (b + 1) * C)
        S = (A[lo:hi] @ A[lo:hi].T).toarray()
        np.fill_diagonal(S, 0)
        part = np.partition(S, S.shape[1] - ntop, axis=1)[:, S.shape[1] - ntop]
        thr[lo:hi] = part
    print("own-block thresholds: mean %.3f, zero rows %d" % (thr.mean(), int((thr == 0).sum())))
    grp = lane_groups()
    rng = np.random.default_rng(0)
    pairs = []
    while len(pairs) < n_pairs:
        a, b = sorted(rng.integers(0, nb - 1, 2))     # (the last, partial block left out)
        if a != b:
            pairs.append((a, b))
    res = {}

    def account(name, hit_cells, grp_of_col):
        # hit_cells: bool [rows, C] -- the cells that are handed over; a step is hit when any of its cells is
        steps = np.zeros((hit_cells.shape[0], 4), bool)
        for t in range(4):
            steps[:, t] = hit_cells[:, (grp_of_col // 64) == t].any(axis=1)
        r = res.setdefault(name, [0, 0, 0])
        r[0] += steps.sum()
        r[1] += steps.size
        r[2] += hit_cells.sum()

    for (bl, bh) in pairs:
        S = (A[bl * C:(bl + 1) * C] @ A[bh * C:(bh + 1) * C].T).toarray()      # from-rows of the lower block x to-rows of the higher
        th = thr[bh * C:(bh + 1) * C]
        # today: exact per-row (16-bit upper halves, conservative: modelled as exact)
        account("as they lie, per-row thresholds", S > th[None, :], grp)
        # group minimum, rows as they lie: prefilter hit = any cell of the lane's eight above the group minimum
        gmin = np.full(256, np.inf, np.float32)
        np.minimum.at(gmin, grp, th)
        account("as they lie, group minimum (all cells above it handed over)", S > gmin[grp][None, :], grp)
        # full sort: slot order = threshold order, groups of eight consecutive
        order = np.argsort(th, kind="stable")
        slot_of_row = np.empty(C, np.int64)
        # group g takes sorted rows 8g .. 8g+7; which slots a group owns does not matter for the counts
        g_of_row = np.empty(C, np.int64)
        g_of_row[order] = np.arange(C) // 8
        gmin = np.full(256, np.inf, np.float32)
        np.minimum.at(gmin, g_of_row, th)
        account("full sort, group minimum", S > gmin[g_of_row][None, :], g_of_row)
        # bank-class sort: row r may only move to slots = r mod 32; class rho's 64 rows sorted, pair i goes to group i of
        # the 32 groups that own slots of the class (groups (t, l) with l mod 8 == rho // 4)
        g_of_row = np.empty(C, np.int64)
        for rho in range(32):
            rows = np.arange(rho, C, 32)
            o = rows[np.argsort(th[rows], kind="stable")]
            m = rho // 4
            gl = [t * 64 + l for t in range(4) for l in range(m, 64, 8)]     # 32 groups
            for i in range(32):
                g_of_row[o[2 * i]] = gl[i]
                g_of_row[o[2 * i + 1]] = gl[i]
        gmin = np.full(256, np.inf, np.float32)
        np.minimum.at(gmin, g_of_row, th)
        account("bank-class sort, group minimum", S > gmin[g_of_row][None, :], g_of_row)
        account("bank-class sort, per-row thresholds in the rare path", S > th[None, :], g_of_row)
    rows = n_pairs * C
    for k, (h, tot, c) in res.items():
        print("%-62s hit steps %5.1f %%   handed over per from-row and block %6.3f  (x ~24 blocks = %5.1f per row)"
              % (k, 100.0 * h / tot, c / rows, 24.0 * c / rows))


if __name__ == "__main__":
    main()
