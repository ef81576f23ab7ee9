"""K7's per-pair arithmetic (polyfuzz_amd/csrc/k7_core.h) on the CPU: the header the HIP kernel calls from its lanes,
compiled with g++ (tests/k7_core_host.cpp), against oracle/fuzz_scorers.c on every pair of seeded lists:
  * the exact score, bit for bit, in every mode and word class;
  * with a running best `cur`: exact whenever the true score reaches cur, never above the true score otherwise (that is
    what lets process.extractOne's maximum skip components);
  * the float32 upper bound the kernel prunes with: bound + slack >= the true score, always.
PARITY UNPINNED beyond the oracle (rapidfuzz is not installable)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests import k7_prep

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
MODES = ["WRatio", "partial_ratio", "token_set_ratio", "token_ratio", "partial_token_sort_ratio", "partial_token_set_ratio",
         "partial_token_ratio"]
SLACK = 0.05          # the kernel prunes a pair only when bound + SLACK < cur


@pytest.fixture(scope="module")
def host():
    so = os.path.join(REPO, "oracle", "_build", "k7_core_host.so")
    src = [os.path.join(HERE, "k7_core_host.cpp"), os.path.join(REPO, "polyfuzz_amd", "csrc", "k7_core.h")]
    os.makedirs(os.path.dirname(so), exist_ok=True)
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", src[0], "-o", so])
    lib = ctypes.CDLL(so)
    lib.k7_host_pairs.restype = ctypes.c_int
    return lib


def run_pairs(lib, W, alpha, A, B, mode, cur, share=0, presence=False):
    lib.k7_host_set_share(share)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    lib.k7_host_set_presence(vp(A["pres"]) if presence else None, vp(B["pres"]) if presence else None)
    out = np.empty((A["n"], B["n"]), np.float64)
    ub = np.empty((A["n"], B["n"]), np.float32)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)

    def args(L):
        return [ctypes.c_int64(L["n"]), p(L["sym"][0]), p(L["off"][0]), p(L["sym"][1]), p(L["off"][1]), p(L["sym"][2]), p(L["off"][2]),
                p(L["tag"]), p(L["tok_off"]), p(L["tok_id"]), p(L["tok_len"]), p(L["hist"]), p(L["usum"])]
    cur = np.ascontiguousarray(cur, np.float64)
    rc = lib.k7_host_pairs(W, alpha.n_sym, *args(A), *args(B), MODES.index(mode), p(cur), p(out), p(ub))
    assert rc == 0
    return out, ub


def _lists(seed, n_from, n_to, long_words=False):
    rng = np.random.default_rng(seed)
    words = ["new", "york", "mets", "braves", "the", "atlanta", "vs", "a", "bb", "inc", "llc", "co", "yankees", "red", "sox",
             "x", "of", "los", "angeles", "dodgers"]
    if long_words:
        words += ["internationalisation", "incorporated", "pharmaceuticals", "telecommunications"]

    def mk(n):
        out = []
        for _ in range(n):
            k = int(rng.integers(0, 7 if not long_words else 10))
            s = (" " * int(rng.integers(1, 3))).join(rng.choice(words, size=k))
            if rng.random() < 0.15:
                s = " " + s + "\t "
            if rng.random() < 0.2 and s:
                q = int(rng.integers(0, len(s)))
                s = s[:q] + "q" + s[q + 1:]
            out.append(s)
        return out
    return mk(n_from), mk(n_to)


CASES = [(1, 3, False), (2, 9, True), (4, 5, True)]


@pytest.mark.parametrize("W,seed,long_words", CASES)
@pytest.mark.parametrize("mode", MODES)
def test_score_and_bound_vs_oracle(host, oracle_mod, mode, W, seed, long_words):
    from polyfuzz_amd import datasets
    fl, tl = _lists(seed, 40, 90, long_words)
    titles_f, titles_t = datasets.c3_lists(60)
    # (strings of spaces: not empty, yet without a token -- their sorted and distinct-token forms ARE empty)
    fl += titles_f[:25] + ["this is a test", "fuzzy was a bear", "", "a", "mets mets mets new", "zzz", "  ", " "]
    tl += titles_t[:50] + ["this is a new test!!!", "fuzzy fuzzy was a bear", "", "this is a test!", "new mets", "a\tb  a", " ", "   ", "\t "]
    if W == 1:
        fl = [s for s in fl if len(s) <= 64]
    fl = [s for s in fl if len(s) <= 64 * W and len(set(s.split())) <= 32]
    alpha = k7_prep.Alphabet(tl)
    A, B = k7_prep.prepare(fl, alpha, False), k7_prep.prepare(tl, alpha, True)
    truth = oracle_mod.fuzz_matrix(fl, tl, mode)
    score, ub = run_pairs(host, W, alpha, A, B, mode, np.zeros(len(fl)))
    np.testing.assert_array_equal(score, truth)
    assert (ub > -999).all(), "a refined bound exceeded the coarse one"
    assert (ub + SLACK >= truth).all(), (mode, float((truth - ub).max()))
    # a running best: the 80th percentile of the row (some pairs above, most below)
    cur = np.quantile(truth, 0.8, axis=1)
    score2, _ = run_pairs(host, W, alpha, A, B, mode, cur)
    reach = truth >= cur[:, None]
    np.testing.assert_array_equal(score2[reach], truth[reach])
    assert (score2 <= truth).all()
    # ... and a running best that IS a score of the row (ties count: extractOne keeps the first of equal scores): the row's
    # best, and its 10th best -- the window sweeps step over windows that cannot reach it, equality must survive
    for cur in (truth.max(axis=1), np.sort(truth, axis=1)[:, -10]):
        score3, _ = run_pairs(host, W, alpha, A, B, mode, cur)
        reach = truth >= cur[:, None]
        np.testing.assert_array_equal(score3[reach], truth[reach])
        assert (score3 <= truth).all()
    # the kernel's way: window sweeps deferred and shared out in runs of windows, each run on its own
    for share in (5, 16 * W):
        score4, _ = run_pairs(host, W, alpha, A, B, mode, np.zeros(len(fl)), share)
        np.testing.assert_array_equal(score4, truth)
        for cur in (truth.max(axis=1), np.quantile(truth, 0.8, axis=1)):
            score5, _ = run_pairs(host, W, alpha, A, B, mode, cur, share)
            reach = truth >= cur[:, None]
            np.testing.assert_array_equal(score5[reach], truth[reach])
            assert (score5 <= truth).all()


def test_bound_is_useful(host, oracle_mod):
    """not only valid: with the row's true best as cur, the bound prunes nearly every pair of real title lists"""
    from polyfuzz_amd import datasets
    fl, tl = datasets.c3_lists(4000)
    fl = [s for s in fl if len(s) <= 64 and len(set(s.split())) <= 32][:40]
    alpha = k7_prep.Alphabet(tl)
    A, B = k7_prep.prepare(fl, alpha, False), k7_prep.prepare(tl, alpha, True)
    truth = oracle_mod.fuzz_matrix(fl, tl, "WRatio")
    _, ub = run_pairs(host, 1, alpha, A, B, "WRatio", np.zeros(len(fl)))
    assert (ub + SLACK >= truth).all()
    kept = (ub + SLACK >= truth.max(axis=1)[:, None]).mean()
    assert kept < 0.15, kept


@pytest.mark.parametrize("W", [1, 2])
def test_window_sweeps_on_a_three_letter_alphabet(host, oracle_mod, W):
    """partial_ratio where windows matter most: strings over {a, b, c} (high LCS everywhere, ties everywhere), all length
    combinations from 1 to 20 (to 70 for two words) -- both window families, equal lengths (both at once), prefixes and
    shrinking suffix windows -- swept in one piece and in shares of 3, 7 and 16 windows, with no floor, with the row's
    best score and with its median as the running best: exact wherever the true score reaches it, never above it."""
    rng = np.random.default_rng(17 + W)
    hi = 20 if W == 1 else 70
    mk = lambda n: ["".join(rng.choice(list("abc"), size=int(rng.integers(1, hi + 1)))) for _ in range(n)]
    fl, tl = mk(45), mk(120)
    fl += ["a", "abcabcabcabc", "c" * 9]
    tl += ["a", "b", "abc" * 6, "cab" * 5 + "a", "c" * 9, "c" * 10]
    alpha = k7_prep.Alphabet(tl)
    A, B = k7_prep.prepare(fl, alpha, False), k7_prep.prepare(tl, alpha, True)
    for mode in ("partial_ratio", "WRatio"):
        truth = oracle_mod.fuzz_matrix(fl, tl, mode)
        for share in (0, 3, 7, 16):
            score, _ = run_pairs(host, W, alpha, A, B, mode, np.zeros(len(fl)), share)
            np.testing.assert_array_equal(score, truth, err_msg=f"{mode} share {share}")
            for cur in (truth.max(axis=1), np.median(truth, axis=1)):
                s2, _ = run_pairs(host, W, alpha, A, B, mode, cur, share)
                reach = truth >= cur[:, None]
                np.testing.assert_array_equal(s2[reach], truth[reach], err_msg=f"{mode} share {share}")
                assert (s2 <= truth).all()


@pytest.mark.parametrize("W,seed,long_words", CASES)
@pytest.mark.parametrize("mode", MODES)
def test_symbol_presence_bound_is_valid(host, oracle_mod, mode, W, seed, long_words):
    """the bound with the symbol-presence term (k7_core.h: fz_presence_miss -- on the CPU only so far): still an upper bound
    of the true score in every mode and word class, never above the bound without the term"""
    fl, tl = _lists(seed + 50, 40, 90, long_words)
    fl += ["this is a test", "fuzzy was a bear", "", "a", "zzz", "qq rr", "new  york"]
    tl += ["this is a new test!!!", "fuzzy fuzzy was a bear", "", "this is a test!", "new mets", "a\tb  a", "york new"]
    fl = [s for s in fl if len(s) <= 64 * W and len(set(s.split())) <= 32]
    alpha = k7_prep.Alphabet(tl)
    A, B = k7_prep.prepare(fl, alpha, False), k7_prep.prepare(tl, alpha, True)
    truth = oracle_mod.fuzz_matrix(fl, tl, mode)
    _, ub0 = run_pairs(host, W, alpha, A, B, mode, np.zeros(len(fl)))
    _, ub1 = run_pairs(host, W, alpha, A, B, mode, np.zeros(len(fl)), presence=True)
    assert (ub1 > -999).all()
    assert (ub1 + SLACK >= truth).all(), (mode, float((truth - ub1).max()))
    assert (ub1 <= ub0 + 1e-4).all()


def test_symbol_presence_bound_is_useful(host, oracle_mod):
    """what the term is worth on real titles: the pairs that reach the row's true best score, with and without it"""
    from polyfuzz_amd import datasets
    fl, tl = datasets.c3_lists(4000)
    fl = [s for s in fl if len(s) <= 64 and len(set(s.split())) <= 32][:40]
    alpha = k7_prep.Alphabet(tl)
    A, B = k7_prep.prepare(fl, alpha, False), k7_prep.prepare(tl, alpha, True)
    for mode in ("WRatio", "partial_ratio"):
        truth = oracle_mod.fuzz_matrix(fl, tl, mode)
        best = truth.max(axis=1)[:, None]
        _, ub0 = run_pairs(host, 1, alpha, A, B, mode, np.zeros(len(fl)))
        _, ub1 = run_pairs(host, 1, alpha, A, B, mode, np.zeros(len(fl)), presence=True)
        assert (ub1 + SLACK >= truth).all()
        k0, k1 = (ub0 + SLACK >= best).mean(), (ub1 + SLACK >= best).mean()
        print(f"{mode}: pairs reaching the row's best: {k0:.4f} -> {k1:.4f} with symbol presence")
        assert k1 < 0.85 * k0, (mode, k0, k1)


@pytest.mark.parametrize("W,seed,long_words", CASES[:2])
@pytest.mark.parametrize("mode", ["WRatio", "partial_ratio", "partial_token_ratio"])
def test_window_sweeps_with_live_masks(host, oracle_mod, mode, W, seed, long_words):
    """an experiment for the next round (FuzzSweep::has_live, not in the kernel): a window that moves on gains matches only
    from entering positions whose symbol the other string has at all -- exact as before, and fewer windows swept"""
    host.k7_host_windows.restype = ctypes.c_longlong
    fl, tl = _lists(seed + 7, 40, 90, long_words)
    fl += ["this is a test", "abc", "a", "mets mets mets new"]
    tl += ["this is a new test!!!", "cab", "a", "new mets", "abcabcabc"]
    fl = [s for s in fl if len(s) <= 64 * W and len(set(s.split())) <= 32]
    alpha = k7_prep.Alphabet(tl)
    A, B = k7_prep.prepare(fl, alpha, False), k7_prep.prepare(tl, alpha, True)
    truth = oracle_mod.fuzz_matrix(fl, tl, mode)
    swept = {}
    try:
        for live in (0, 1):
            host.k7_host_set_live(live)
            host.k7_host_windows()
            for cur in (np.zeros(len(fl)), truth.max(axis=1), np.quantile(truth, 0.8, axis=1)):
                score, _ = run_pairs(host, W, alpha, A, B, mode, cur, share=16)
                reach = truth >= cur[:, None]
                np.testing.assert_array_equal(score[reach], truth[reach])
                assert (score <= truth).all()
            swept[live] = host.k7_host_windows()
    finally:
        host.k7_host_set_live(0)
    assert swept[1] <= swept[0]
