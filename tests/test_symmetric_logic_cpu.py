"""The argument behind K3's symmetric form (polyfuzz_amd/csrc/k3_symmetric.hip), in executable form on the CPU.

The kernel itself only runs on the GPU (tests/test_k3_cossim_gpu.py::test_symmetric_* hold it to the row-major kernel bit
for bit).  What can be held HERE is what it relies on:

1. K3's fixed-point sums are symmetric bit for bit: trunc(fl32(fl32(a * S) * b)) == trunc(fl32(fl32(b * S) * a)) for S a power
   of two, so s(i, j) summed in int32 equals s(j, i) -- a pair may be scored by either row.
2. The three passes + merge select exactly the top-n of the row-major rule -- every sum > thr0, ordered by (sum desc, column
   asc) -- although a row only looks at its own block and the blocks above, thresholds are taken from the own-block pass,
   candidates are handed over on the UPPER 16 BITS of sums and thresholds (a conservative test), a row's push slots overflow
   (the row is recomputed in full), and a row's own threshold keeps rising while it is sent candidates.

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
    thr16 = thr >> 16                                       # what the pushers see of a row's threshold
    # pass 1: row j x the blocks above; own candidates by the row's running threshold, foreign ones by the cells' rows'
    pushed = [[] for _ in range(n)]
    for j in range(n):
        keys, t = list(own[j]), thr[j]
        for b in range(block[j] + 1, nb):
            for i in range(b * C, min(n, (b + 1) * C)):
                s = S[j, i]
                if s > t:
                    keys.append(_key(s, i))
                    if len(keys) > ntop + 7:                # a compaction now and then: the threshold rises mid-row
                        keys = _top_n(keys, ntop)
                        t = max(t, thr_of(keys))
                if (s >> 16) >= thr16[i] and s > thr0:      # conservative: upper halves only
                    pushed[i].append(_key(s, j))
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
    # what is handed over is never more than what the exact thresholds would let through plus the 16-bit slack
    assert all(len(set(p)) == len(p) for p in pushed)       # a pair is handed over once
