"""GPU parity of K4 (all-pairs Indel ratio + first arg-max) against the oracle's plain-DP LCS."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _matrix(ctx, fl, tl):
    from polyfuzz_amd import _lib
    f = _lib.DeviceStrings.upload(ctx, fl)
    t = _lib.DeviceStrings.upload(ctx, tl)
    return _lib.indel_matrix(ctx, f, t), _lib.indel_argmax(ctx, f, t)


def test_readme_matrix_known_answers(ctx):
    fl = ["apple", "apples", "appl", "recal", "house", "similarity"]
    tl = ["apple", "apples", "mouse"]
    m, (idx, score) = _matrix(ctx, fl, tl)
    exp = np.array([[100, 90.909091, 20], [90.909091, 100, 18.181818], [88.888889, 80, 0], [40, 36.363636, 20],
                    [20, 18.181818, 80], [13.333333, 12.5, 13.333333]])
    np.testing.assert_allclose(m, exp, atol=1e-6)
    np.testing.assert_array_equal(idx, [0, 1, 0, 0, 2, 0])       # first maximum wins (rows 3 and 5 tie)


def test_published_rapidfuzz_values(ctx):
    """The values rapidfuzz itself publishes for fuzz.ratio / Indel (README and API docs), on the device."""
    m, _ = _matrix(ctx, ["this is a test", "lewenstein", "", "abc"], ["this is a test!", "levenshtein", ""])
    assert m[0, 0] == 96.55172413793103
    assert abs(m[1, 1] - 85.71428571428572) < 1e-12
    assert m[2, 2] == 100.0 and m[3, 2] == 0.0 and m[2, 0] == 0.0


def test_third_party_lcs_pin(ctx, golden):
    """K4 against the third-party fixture directly (textdistance.lcsseq / nltk.edit_distance, make_golden_lcs.py): the
    diagonal of the all-pairs matrix of the fixture's 3 029 pairs, no oracle in between."""
    pairs = golden["lcs_golden"]["pairs"]
    fl, tl = [p[0] for p in pairs], [p[1] for p in pairs]
    m, _ = _matrix(ctx, fl, tl)
    lensum = np.array([len(a) + len(b) for a, b in zip(fl, tl)], np.float64)
    dist = np.array([p[3] for p in pairs], np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        expect = np.where(lensum > 0, (1.0 - dist / lensum) * 100, 100.0)
    np.testing.assert_array_equal(np.diagonal(m), expect)


def test_titles_bit_exact_vs_oracle(ctx, oracle_mod, golden):
    t = golden["titles_lists"]
    fl, tl = t["from_list"], t["to_list"]
    m, (idx, score) = _matrix(ctx, fl, tl)
    e_idx, e_score, e_m = oracle_mod.indel_argmax(fl, tl, want_matrix=True)
    np.testing.assert_array_equal(m, e_m)            # float64, same formula: bit-exact
    np.testing.assert_array_equal(idx, e_idx)
    np.testing.assert_array_equal(score, e_score)
    np.testing.assert_array_equal(idx, golden["npz"]["titles_idx_norm0"])
    np.testing.assert_array_equal(score, golden["npz"]["titles_sim_norm0"])


def test_word_classes_and_edge_lengths(ctx, oracle_mod):
    rng = np.random.default_rng(11)
    alpha = "abcdefgh "
    lens_f = [0, 1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 200, 256, 257, 400, 512, 700, 1024]
    lens_t = [0, 1, 3, 17, 32, 64, 100, 130, 300, 600, 1500] + list(rng.integers(1, 40, 120))
    mk = lambda n: "".join(alpha[i] for i in rng.integers(0, len(alpha), n))
    fl, tl = [mk(n) for n in lens_f], [mk(int(n)) for n in lens_t]
    m, (idx, score) = _matrix(ctx, fl, tl)
    e_idx, e_score, e_m = oracle_mod.indel_argmax(fl, tl, want_matrix=True)
    np.testing.assert_array_equal(m, e_m)
    np.testing.assert_array_equal(idx, e_idx)
    np.testing.assert_array_equal(score, e_score)
    assert m[0, 0] == 100.0          # both empty


def test_wide_alphabet_16bit_symbols(ctx, oracle_mod):
    rng = np.random.default_rng(5)
    cps = list(range(0x4E00, 0x4E00 + 400)) + list(range(0x41, 0x5B))     # > 255 distinct symbols
    mk = lambda n: "".join(chr(cps[i]) for i in rng.integers(0, len(cps), n))
    fl = [mk(int(n)) for n in rng.integers(0, 50, 40)]
    tl = [mk(int(n)) for n in rng.integers(0, 50, 150)] + fl[:5]
    m, (idx, score) = _matrix(ctx, fl, tl)
    e_idx, e_score, e_m = oracle_mod.indel_argmax(fl, tl, want_matrix=True)
    np.testing.assert_array_equal(m, e_m)
    np.testing.assert_array_equal(idx, e_idx)


def test_self_match_removes_first_equal(ctx, oracle_mod, golden):
    from polyfuzz_amd import _lib
    dup = golden["titles_self_list"]["from_list"]
    first = {}
    for j, s in enumerate(dup):
        first.setdefault(s, j)
    skip = np.array([first[s] for s in dup], np.int32)
    f = _lib.DeviceStrings.upload(ctx, dup)
    idx, score = _lib.indel_argmax(ctx, f, f, skip)
    e_idx, e_score = oracle_mod.indel_argmax(dup, dup, self_match=True)
    np.testing.assert_array_equal(idx, e_idx)
    np.testing.assert_array_equal(score, e_score)
    np.testing.assert_array_equal(score, golden["npz"]["titles_self_sim"])
    assert [dup[j] for j in idx] == list(golden["npz"]["titles_self_to"])
    # row shard
    idx2, score2 = _lib.indel_argmax(ctx, f, f, skip, 40, 90)
    np.testing.assert_array_equal(idx2, idx[40:90])


def test_general_kernel_long_strings_and_forced(ctx, oracle_mod, monkeypatch):
    """From-strings beyond 1024 characters (and alphabets whose match table does not fit LDS) take the general kernel:
    any number of 64-bit words, match table and V columns in global scratch.  Bit-exact against the oracle; forcing
    EVERY row through it (PFZ_K4_FORCE_GENERAL) reproduces the word-class kernels' results."""
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(12)
    alpha = np.array(list("abcdefghijklmnop qrst"), dtype=object)

    def mk(n, lo, hi):
        return ["".join(rng.choice(alpha, size=int(rng.integers(lo, hi))).tolist()) for _ in range(n)]
    fl = mk(6, 1025, 2600) + ["a" * 1025, "ab" * 2000] + mk(40, 0, 200)
    tl = mk(150, 0, 300) + mk(5, 900, 3000) + ["", "a" * 1025]
    f, t = _lib.DeviceStrings.upload(ctx, fl), _lib.DeviceStrings.upload(ctx, tl)
    idx, score = _lib.indel_argmax(ctx, f, t)
    e_idx, e_score = oracle_mod.indel_argmax(fl, tl)
    np.testing.assert_array_equal(idx, e_idx)
    np.testing.assert_array_equal(score, e_score)
    m = _lib.indel_matrix(ctx, f, t, 0, 3)
    _, _, e_m = oracle_mod.indel_argmax(fl, tl, rows=(0, 3), want_matrix=True)
    np.testing.assert_array_equal(m, e_m)
    monkeypatch.setenv("PFZ_K4_FORCE_GENERAL", "1")
    idx2, score2 = _lib.indel_argmax(ctx, f, t)
    np.testing.assert_array_equal(idx2, e_idx)
    np.testing.assert_array_equal(score2, e_score)


def test_rapidfuzz_matcher_extract_one_rules(oracle_mod):
    """RapidFuzz(scorer='ratio'): process.extractOne semantics on top of K4 -- first best choice, None / 0.0 below
    score_cutoff (a fraction of 1, compared on the 0..100 scale), Similarity = score / 100, a self-match excludes
    only the from-string's own first occurrence and does NOT shrink the list (reference _rapidfuzz.py:99-113 with
    its in-place `to_list.remove` fixed).  Scorer parity is unpinned (rapidfuzz is not installable): the expected
    scores come from oracle/indel.c."""
    from polyfuzz_amd.models import RapidFuzz
    fl = ["apple", "apples", "appl", "recal", "house", "similarity", ""]
    tl = ["apple", "apples", "mouse", ""]
    with pytest.raises(NotImplementedError):
        RapidFuzz(scorer=len)                             # an arbitrary callable has no kernel (and there is no CPU path)
    for cutoff in (0, 0.5, 0.95):
        df = RapidFuzz(scorer="ratio", score_cutoff=cutoff).match(fl, tl)
        assert list(df.columns) == ["From", "To", "Similarity"] and len(df) == len(fl)
        o_idx, o_score = oracle_mod.indel_argmax(fl, tl)
        exp_to = [tl[j] if s >= cutoff * 100 else None for j, s in zip(o_idx, o_score)]
        exp_sim = [s / 100 if s >= cutoff * 100 else 0.0 for s in o_score]
        assert df["To"].tolist() == exp_to and df["Similarity"].tolist() == exp_sim
    df = RapidFuzz(scorer="ratio").match(fl, tl)
    assert df["To"].tolist()[:3] == ["apple", "apples", "apple"] and df["Similarity"].tolist()[-1] == 1.0   # "" vs "": 100
    # token_sort_ratio = ratio of the sorted, single-space-joined tokens; the choice returned is the ORIGINAL string
    fl2 = ["new york mets", "mets  york new", "atlanta braves", "", "braves"]
    tl2 = ["york new mets", "new york yankees", "braves atlanta", "the braves", ""]
    d2 = RapidFuzz(scorer="token_sort_ratio").match(fl2, tl2)
    srt = lambda l: [" ".join(sorted(s.split())) for s in l]
    o_idx, o_score = oracle_mod.indel_argmax(srt(fl2), srt(tl2))
    assert d2["To"].tolist() == [tl2[j] for j in o_idx] and d2["Similarity"].tolist() == [s / 100 for s in o_score]
    assert d2["To"].tolist()[:3] == ["york new mets", "york new mets", "braves atlanta"] and d2["Similarity"].tolist()[:3] == [1.0] * 3
    d2 = RapidFuzz(scorer="token_sort_ratio").match(fl2)                   # self-match: own first occurrence skipped
    assert d2["To"].tolist()[:2] == ["mets  york new", "new york mets"] and d2["Similarity"].tolist()[:2] == [1.0, 1.0]
    q = RapidFuzz(scorer="QRatio").match(fl, tl)
    assert q["To"].tolist()[-1] == "apple" and q["Similarity"].tolist()[-1] == 0.0                          # QRatio("", x) = 0
    assert q["To"].tolist()[:-1] == df["To"].tolist()[:-1]
    # self-match: every string keeps all OTHER strings as choices, whatever the row order
    sl = ["apple", "apples", "appl", "apple"]
    df = RapidFuzz(scorer="ratio").match(sl)
    assert df["To"].tolist() == ["apple", "apple", "apple", "apple"]      # row 0 finds the duplicate at index 3, row 3 finds index 0
    assert df["Similarity"].tolist()[0] == 1.0 and df["Similarity"].tolist()[3] == 1.0


def test_quad_kernel_equals_single_string_kernel_and_oracle(ctx, oracle_mod, monkeypatch):
    """From-strings of <= 32 characters go four per workgroup pass (one ds_read_b128 serves four recurrences);
    longer ones take the one-string word classes.  Both give the oracle's arg-max and float64 score bit for bit --
    row counts that are not multiples of four, empty strings on both sides, duplicates, a self-match with skipped
    first occurrences, characters the to-list never uses."""
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(8)
    alpha = np.array(list("abcdefghij XYZ-'é"), dtype=object)

    def mk(n, lo, hi):
        return ["".join(rng.choice(alpha, size=int(rng.integers(lo, hi))).tolist()) for _ in range(n)]
    fl = mk(203, 0, 33) + ["", "q" * 32, "zzzz"] + mk(7, 33, 90)         # class 0 rows (not a multiple of 4) + longer ones
    tl = mk(530, 0, 60) + ["", "abc", "abc"]
    for self_match in (False, True):
        f = _lib.DeviceStrings.upload(ctx, fl)
        t = f if self_match else _lib.DeviceStrings.upload(ctx, tl)
        names = fl if self_match else tl
        skip = None
        if self_match:
            first = {}
            for j, s in enumerate(names):
                first.setdefault(s, j)
            skip = np.array([first[s] for s in fl], np.int32)
        idx, score = _lib.indel_argmax(ctx, f, t, skip)
        monkeypatch.setenv("PFZ_K4_NO_QUAD", "1")
        idx1, score1 = _lib.indel_argmax(ctx, f, t, skip)
        monkeypatch.delenv("PFZ_K4_NO_QUAD")
        np.testing.assert_array_equal(idx, idx1)
        np.testing.assert_array_equal(score, score1)
        e_idx, e_score, mat = oracle_mod.indel_argmax(fl, names, want_matrix=True)
        if self_match:                      # the oracle's self_match drops j == i; the reference drops the FIRST equal element
            for i in range(len(fl)):
                row = mat[i].copy()
                row[skip[i]] = -1.0
                e_idx[i], e_score[i] = int(np.argmax(row)), row.max()
        np.testing.assert_array_equal(idx, e_idx)
        np.testing.assert_array_equal(score, e_score)


@pytest.mark.parametrize("parts", ["1", "3", "7"])
def test_to_groups_split_over_workgroups(ctx, oracle_mod, monkeypatch, parts):
    """A from-string's (or quad's) to-groups can be split over several workgroups whose bests meet in k4_merge_parts
    (few long from-strings would otherwise run as a handful of workgroups).  Any split gives the oracle's first
    arg-max and score bit for bit: ties across parts resolve to the lowest original index, skipped first occurrences
    of a self-match stay skipped, rows with no candidate at all (a one-string self-match) stay -1."""
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(31)
    alpha = np.array(list("abcdefgh ij"), dtype=object)

    def mk(n, lo, hi):
        return ["".join(rng.choice(alpha, size=int(rng.integers(lo, hi))).tolist()) for _ in range(n)]
    fl = mk(50, 0, 33) + mk(9, 33, 64) + mk(5, 65, 200) + ["abc", "abc"]
    tl = mk(700, 0, 40) + ["abc"] * 3 + mk(300, 0, 90) + ["abc"]         # equal best candidates in different groups
    monkeypatch.setenv("PFZ_K4_PARTS", parts)
    f, t = _lib.DeviceStrings.upload(ctx, fl), _lib.DeviceStrings.upload(ctx, tl)
    idx, score = _lib.indel_argmax(ctx, f, t)
    e_idx, e_score = oracle_mod.indel_argmax(fl, tl)
    np.testing.assert_array_equal(idx, e_idx)
    np.testing.assert_array_equal(score, e_score)
    first = {}
    for j, s in enumerate(fl):
        first.setdefault(s, j)
    skip = np.array([first[s] for s in fl], np.int32)
    idx, score = _lib.indel_argmax(ctx, f, f, skip)
    _, _, mat = oracle_mod.indel_argmax(fl, fl, want_matrix=True)
    for i in range(len(fl)):
        row = mat[i].copy()
        row[skip[i]] = -1.0
        assert idx[i] == int(np.argmax(row)) and score[i] == row.max()
    one = _lib.DeviceStrings.upload(ctx, ["solo"])
    idx, score = _lib.indel_argmax(ctx, one, one, np.array([0], np.int32))
    assert idx[0] == -1 and score[0] == 0.0
