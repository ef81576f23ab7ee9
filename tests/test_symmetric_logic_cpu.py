"""The argument behind K3's symmetric form (polyfuzz_amd/csrc/k3_symmetric.hip), in executable form on the CPU.

The kernel itself only runs on the GPU (tests/test_k3_cossim_gpu.py::test_symmetric_* hold it to the row-major kernel bit
for bit).  What can be held HERE is what it relies on:

1. K3's fixed-point sums are symmetric bit for bit: trunc(fl32(fl32(a * S) * b)) == trunc(fl32(fl32(b * S) * a)) for S a power
   of two, so s(i, j) summed in int32 equals s(j, i) -- a pair may be scored by either row.
2. The three passes + merge select exactly the top-n of the row-major rule -- every sum > thr0, ordered by (sum desc, column
   asc) -- although a row only looks at its own block and the blocks above, thresholds are taken from the own-block pass,
   the rows of a block are RE-DEALT to the accumulator slots by threshold and a sweep step is decided by the minimum
   threshold of a lane's eight slots (round 5), candidates are STAGED by slot and resolved later (the row that owns the slot,
   its exact threshold, the from-row's own threshold as it is by then), a row's push slots overflow (the row is recomputed in
   full), and a row's own threshold keeps rising while it is sent candidates.
3. The re-deal itself (sym_slot of k3_symmetric.hip, restated): a bijection of a block's rows onto its slots that keeps every
   row in its LDS bank and gives the eight slots a lane reads in one sweep step rows of neighbouring threshold ranks.

The emulation follows the kernel's rules literally (the same filters, the same keys, the same overflow rule) on a matrix
small enough for brute force; it is test infrastructure, nothing in the package imports it.
"""
import numpy as np
import pytest

from tests.helpers import random_csr


def _fixed_point_sums(a3, n_col, scale):
    """int64 matrix of K3's sums for every (from-row, to-row): per n-gram trunc(fl32(fl32(a * S) * b)), summed as integers"""
    ip, ix, dv = a3
    n = len(ip) - 1
    dense = np.zeros((n, n_col), np.float32)
    for r in range(n):
        dense[r, ix[ip[r]:ip[r + 1]]] = dv[ip[r]:ip[r + 1]].astype(np.float32)
    scaled = (dense * np.float32(scale)).astype(np.float32)            # exact: S is a power of two
    out = np.zeros((n, n), np.int64)
    for k in range(n_col):
        col_s, col = scaled[:, k], dense[:, k]
        nz = np.nonzero(col)[0]
        if len(nz):
            prod = (col_s[nz][:, None] * col[nz][None, :]).astype(np.float32)     # fl32 product, from-row's value scaled
            out[np.ix_(nz, nz)] += np.trunc(prod).astype(np.int64)
    return out


def test_fixed_point_products_are_symmetric_bit_for_bit():
    rng = np.random.default_rng(11)
    a = rng.random(200000, dtype=np.float32) * np.float32(3.7)
    b = rng.random(200000, dtype=np.float32) * np.float32(0.9)
    for k in (30, 27, 12):
        s = np.float32(2.0 ** k)
        lhs = np.trunc(((a * s).astype(np.float32) * b).astype(np.float32))
        rhs = np.trunc(((b * s).astype(np.float32) * a).astype(np.float32))
        assert np.array_equal(lhs, rhs)
    a3 = random_csr(rng, 300, 90, 0.06)
    sums = _fixed_point_sums(a3, 90, 2.0 ** 30)
    assert np.array_equal(sums, sums.T)                 # the whole matrix of integer sums


def _top_n(keys, ntop):
    return sorted(keys, reverse=True)[:ntop]


def _key(s, col):
    return (int(s) << 32) | (~int(col) & 0xFFFFFFFF)


@pytest.mark.parametrize("ntop,thr0,cap", [(5, 0, 6), (1, 0, 3), (8, 1 << 26, 1000), (3, 0, 0)])
def test_three_passes_and_merge_select_the_row_major_top_n(ntop, thr0, cap):
    rng = np.random.default_rng(5 + ntop)
    n, n_col, C = 330, 70, 32                              # 11 blocks of 32 rows, the last one partial
    a3 = random_csr(rng, n, n_col, 0.07, empty_rows=(3, 40, 329))
    ip, ix, dv = [np.array(x) for x in a3]
    rows = [(ix[ip[r]:ip[r + 1]], dv[ip[r]:ip[r + 1]]) for r in range(n)]
    for r in range(45, n, 41):                              # copies of row 7 in several blocks: exact ties, the column decides
        rows[r] = rows[7]
    ptr = np.zeros(n + 1, np.int64)
    for r in range(n):
        ptr[r + 1] = ptr[r] + len(rows[r][0])
    a3 = (ptr, np.concatenate([c for c, _ in rows]).astype(np.int32), np.concatenate([v for _, v in rows]))
    S = _fixed_point_sums(a3, n_col, 2.0 ** 30)
    nb = (n + C - 1) // C
    block = np.arange(n) // C
    # the row-major rule, by brute force
    want = []
    for i in range(n):
        want.append(_top_n([_key(S[i, j], j) for j in range(n) if j != i and S[i, j] > thr0], ntop))

    def thr_of(keys):                                       # compact(): the ntop-th best - 1 once ntop keys are there
        return (keys[ntop - 1] >> 32) - 1 if len(keys) >= ntop else thr0

    # pass 0: every row x its own block
    own = []
    thr = np.zeros(n, np.int64)
    for i in range(n):
        cols = [j for j in range(block[i] * C, min(n, (block[i] + 1) * C)) if j != i and S[i, j] > thr0]
        own.append(_top_n([_key(S[i, j], j) for j in cols], ntop))
        thr[i] = max(thr_of(own[i]), thr0)
    # the re-deal: inside a block the rows are ranked by (threshold, row) and dealt to groups of eight slots (the kernel ranks
    # inside bank classes; any grouping by rank exercises the same rule); gmin = the minimum threshold of a group
    group = np.zeros(n, np.int64)
    gmin = {}
    for b in range(nb):
        members = sorted(range(b * C, min(n, (b + 1) * C)), key=lambda i: (thr[i], i))
        for rank, i in enumerate(members):
            group[i] = b * C + rank // 8
            gmin[group[i]] = min(gmin.get(group[i], 1 << 62), thr[i])
    # pass 1: row j x the blocks above.  The sweep sees slots, not rows: a sum above min(own threshold, gmin of its group) is
    # STAGED with two flags; the stage is drained now and then -- a foreign candidate is pushed when it beats its row's exact
    # threshold, an own candidate is kept when it beats the from-row's threshold as it is at that moment
    pushed = [[] for _ in range(n)]
    for j in range(n):
        keys, t = list(own[j]), thr[j]
        stage = []

        def drain():
            nonlocal keys, t
            for (s, i, f_own, f_fgn) in stage:
                if f_fgn and s > thr[i]:
                    pushed[i].append(_key(s, j))
                if f_own and s > t:
                    keys.append(_key(s, i))
                    if len(keys) > ntop + 7:                # a compaction now and then: the threshold rises mid-row
                        keys = _top_n(keys, ntop)
                        t = max(t, thr_of(keys))
            stage.clear()

        for b in range(block[j] + 1, nb):
            for i in range(b * C, min(n, (b + 1) * C)):
                s = S[j, i]
                g = gmin[group[i]]
                if s > min(t, g):
                    stage.append((s, i, s > t, s > g))
                    if len(stage) > 9:
                        drain()
        drain()
        own[j] = _top_n(keys, ntop)
    # merge, and the rows that were sent more than their slots: recomputed in full
    overflowed = 0
    for i in range(n):
        if len(pushed[i]) > cap:
            overflowed += 1
            got = _top_n([_key(S[i, j], j) for j in range(n) if j != i and S[i, j] > thr0], ntop)
        else:
            got = _top_n(own[i] + pushed[i], ntop)
        assert got == want[i], (i, block[i])
    if cap <= 6:
        assert overflowed > 0                               # the overflow path was exercised
    assert all(len(set(p)) == len(p) for p in pushed)       # a pair is handed over once
    # exactly what the rows' own-block thresholds let through is handed over
    assert sum(len(p) for p in pushed) == sum(1 for j in range(n) for i in range((block[j] + 1) * C, n) if S[j, i] > thr[i])


def _sym_slot(rho, i):
    """sym_slot() of k3_symmetric.hip: bank class rho (row mod 32), rank i among the class' 64 rows -> (slot, step, lane, place)"""
    p, half, c = i >> 1, i & 1, rho & 3
    t, lane = p >> 3, (rho >> 2) + 8 * (p & 7)
    return 512 * t + 256 * half + 4 * lane + c, t, lane, half * 4 + c


def test_slot_re_deal_is_a_bank_preserving_bijection():
    seen = {}
    for rho in range(32):
        for i in range(64):
            slot, t, lane, e = _sym_slot(rho, i)
            assert slot % 32 == rho                              # the row stays in its LDS bank
            assert 0 <= slot < 2048 and slot not in seen
            seen[slot] = (rho, i)
            # what the sweep reads: step t, lane l -> int4 slots 128 t + l and 128 t + l + 64, i.e. cells 512 t + 4 l + c and + 256
            assert slot == 512 * t + 4 * lane + (e & 3) + 256 * (e >> 2) and 0 <= t < 4 and 0 <= lane < 64
    assert len(seen) == 2048
    # a lane-step's eight slots hold ranks 2p, 2p+1 of four neighbouring bank classes: neighbouring threshold ranks
    by_group = {}
    for slot, (rho, i) in seen.items():
        t, rem = divmod(slot, 512)
        by_group.setdefault((t, (rem % 256) >> 2), []).append((rho, i))
    for members in by_group.values():
        assert len(members) == 8
        assert len({i >> 1 for _, i in members}) == 1 and len({rho >> 2 for rho, _ in members}) == 1
