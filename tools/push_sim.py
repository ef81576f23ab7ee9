#!/usr/bin/env python3
"""Design aid: how many candidates does K3's sweep push (and how often does it compact) per from-row?"""
import numpy as np
import scipy.sparse as sp
z = np.load("/tmp/sim/mats.npz")
A = sp.csr_matrix((z["ad"], z["ai"], z["ap"]), shape=tuple(z["ashape"]))
B = sp.csr_matrix((z["bd"], z["bi"], z["bp"]), shape=tuple(z["bshape"]))
n_to = B.shape[0]; C = 2048; nb = (n_to + C - 1) // C; ntop = 5; cap_trigger = 32
rng = np.random.default_rng(0)
rows = rng.choice(A.shape[0], 400, replace=False)
tot_push = tot_comp = 0; first_blk_push = 0; warm_push = 0
for i in rows:
    full = np.asarray((B @ A[i].T).todense()).ravel()
    thr = 0.0; cand = []; pushes = comps = 0
    for b in range(nb):
        blk = full[b * C:(b + 1) * C]
        # sweep steps of 512 columns, 8 sub-columns each: the threshold is re-read per sub-column; approximate per step
        for s0 in range(0, len(blk), 512):
            v = blk[s0:s0 + 512]
            sel = v[v > thr]
            if len(sel):
                cand.extend(sel.tolist()); pushes += len(sel)
                if b == 0: first_blk_push += len(sel)
                while len(cand) > cap_trigger:
                    cand = sorted(cand, reverse=True)[:ntop]; comps += 1
                    if len(cand) == ntop: thr = max(thr, cand[-1] - 1e-12)
    tot_push += pushes; tot_comp += comps
    # warm start: n-th largest of the 64 lane maxima of block 0
    blk0 = full[:C].reshape(-1, 64)            # column j of a 64-wide layout ~ lane
    lane_max = blk0.max(axis=0)
    t0 = np.sort(lane_max)[::-1][ntop - 1]
    warm_push += int((full[:C] > max(t0 - 1e-12, 0)).sum())
print(f"pushes/row {tot_push/len(rows):.0f}, compactions/row {tot_comp/len(rows):.1f}, pushes in block 0: {first_blk_push/len(rows):.0f}; with warm-started threshold block 0 pushes {warm_push/len(rows):.0f}")
