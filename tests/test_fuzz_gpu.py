"""GPU parity of K7 (the per-pair rapidfuzz scorers: WRatio, partial_ratio, token_set_ratio, token_ratio,
partial_token_*_ratio) against oracle/fuzz_scorers.py -- first best choice and float64 score, bit for bit.
PARITY UNPINNED beyond the published values the oracle is anchored on (tests/test_fuzz_oracle_cpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODES = ["WRatio", "partial_ratio", "token_set_ratio", "token_ratio", "partial_token_sort_ratio",
         "partial_token_set_ratio", "partial_token_ratio"]


def _lists(seed, n_from, n_to, long_words=False):
    rng = np.random.default_rng(seed)
    words = ["new", "york", "mets", "braves", "the", "atlanta", "vs", "a", "bb", "inc", "llc", "co", "yankees", "red", "sox",
             "x", "of", "los", "angeles", "dodgers"]
    if long_words:
        words += ["internationalisation", "incorporated", "pharmaceuticals", "telecommunications"]

    def mk(n):
        out = []
        for _ in range(n):
            k = int(rng.integers(0, 7))
            toks = list(rng.choice(words, size=k))
            s = (" " * int(rng.integers(1, 3))).join(toks)
            if rng.random() < 0.15:
                s = " " + s + "  "
            if rng.random() < 0.2 and s:
                p = int(rng.integers(0, len(s)))
                s = s[:p] + "q" + s[p + 1:]
            out.append(s)
        return out
    return mk(n_from), mk(n_to)


@pytest.mark.parametrize("mode", MODES)
def test_modes_bit_exact_vs_oracle(ctx, mode):
    from oracle import fuzz_scorers as f
    from polyfuzz_amd import _lib
    fl, tl = _lists(3, 70, 150)
    fl += ["this is a test", "this is a word", "fuzzy was a bear", "", "a", "mets mets mets new"]
    tl += ["this is a new test!!!", "THIS IS A WORD", "fuzzy fuzzy was a bear", "", "this is a test!", "new mets"]
    idx, score = _lib.fuzz_extract_one(ctx, fl, tl, mode)
    e_idx, e_score = f.extract_one_all(fl, tl, f.SCORERS[mode])
    np.testing.assert_array_equal(score, np.array(e_score))
    np.testing.assert_array_equal(idx, np.array(e_idx, np.int32))


def test_published_values_on_the_device(ctx):
    from polyfuzz_amd import _lib
    one = lambda a, b, m: float(_lib.fuzz_extract_one(ctx, [a], [b], m)[1][0])
    assert one("this is a test", "this is a new test!!!", "WRatio") == 85.5
    assert abs(one("this is a word", "THIS IS A WORD", "WRatio") - 21.42857142857143) < 1e-12
    assert one("this is a test", "this is a test!", "partial_ratio") == 100.0
    assert one("fuzzy was a bear", "fuzzy fuzzy was a bear", "token_set_ratio") == 100.0
    assert abs(one("fuzzy was a bear", "fuzzy fuzzy was a bear", "token_ratio") - 100.0) < 1e-12


def test_two_word_strings_self_match_and_parts(ctx, monkeypatch):
    """From-strings between 65 and 128 characters (two 64-bit words), a self-match (own first occurrence skipped,
    duplicates find each other), the to-groups split over several workgroups, one query against many choices."""
    from oracle import fuzz_scorers as f
    from polyfuzz_amd import _lib
    fl, _ = _lists(9, 60, 1, long_words=True)
    fl += [fl[3], fl[10], "internationalisation pharmaceuticals telecommunications incorporated of los angeles",
           "telecommunications incorporated  of new york and pharmaceuticals internationalisation inc",
           "the internationalisation of telecommunications incorporated the the pharmaceuticals of atlanta braves vs red sox x"]
    assert max(map(len, fl)) > 64
    first = {}
    for j, s in enumerate(fl):
        first.setdefault(s, j)
    skip = np.array([first[s] for s in fl], np.int32)
    for parts in ("1", "3"):
        monkeypatch.setenv("PFZ_K7_PARTS", parts)
        for mode in ("WRatio", "token_ratio", "partial_ratio"):
            idx, score = _lib.fuzz_extract_one(ctx, fl, fl, mode, skip)
            e_idx, e_score = f.extract_one_all(fl, fl, f.SCORERS[mode], skip)
            np.testing.assert_array_equal(score, np.array(e_score))
            np.testing.assert_array_equal(idx, np.array(e_idx, np.int32))
    monkeypatch.delenv("PFZ_K7_PARTS")
    q, choices = ["los angeles dodgers inc"], _lists(4, 1, 700)[1]
    idx, score = _lib.fuzz_extract_one(ctx, q, choices, "WRatio")
    e_idx, e_score = f.extract_one_all(q, choices, f.WRatio)
    assert idx[0] == e_idx[0] and score[0] == e_score[0]


def test_strings_beyond_the_fast_kernel(ctx, oracle_mod, monkeypatch):
    """No loud limit any more: a 400-character from-string, one of 40 distinct tokens, a to-string of 35 distinct tokens,
    and -- PFZ_K7_FORCE_GENERAL -- ordinary strings through the general kernel: every scorer, bit for bit."""
    from polyfuzz_amd import _lib
    fl, tl = _lists(11, 12, 60, long_words=True)
    rng = np.random.default_rng(5)
    words = ["alpha", "beta", "gamma", "delta", "of", "the", "inc", "llc", "new", "york", "mets", "braves"]
    long_from = " ".join(rng.choice(words, size=70))                       # ~ 400 characters, few distinct tokens
    many_tok = " ".join(f"t{i}" for i in range(40))                        # 40 distinct tokens
    fl += [long_from[:400], many_tok, "new york mets", ""]
    tl += [" ".join(f"t{i}" for i in range(3, 38)), long_from[5:300], "york new mets", "t1 t2 t3", ""]
    assert max(map(len, fl)) > 256 and max(len(set(s.split())) for s in tl) > 32
    for mode in MODES:
        idx, score = _lib.fuzz_extract_one(ctx, fl, tl, mode)
        e_idx, e_score = oracle_mod.fuzz_extract_one(fl, tl, mode)
        np.testing.assert_array_equal(score, e_score, err_msg=mode)
        np.testing.assert_array_equal(idx, e_idx, err_msg=mode)
    monkeypatch.setenv("PFZ_K7_FORCE_GENERAL", "1")
    fl2, tl2 = _lists(12, 25, 40)
    first = {}
    for j, s in enumerate(fl2):
        first.setdefault(s, j)
    skip = np.array([first[s] for s in fl2], np.int32)
    for mode in MODES:
        idx, score = _lib.fuzz_extract_one(ctx, fl2, tl2, mode)
        e_idx, e_score = oracle_mod.fuzz_extract_one(fl2, tl2, mode)
        np.testing.assert_array_equal(score, e_score, err_msg=mode)
        np.testing.assert_array_equal(idx, e_idx, err_msg=mode)
    idx, score = _lib.fuzz_extract_one(ctx, fl2, fl2, "WRatio", skip)
    e_idx, e_score = oracle_mod.fuzz_extract_one(fl2, fl2, "WRatio", skip)
    np.testing.assert_array_equal(score, e_score)
    np.testing.assert_array_equal(idx, e_idx)
    monkeypatch.delenv("PFZ_K7_FORCE_GENERAL")
    idx, score = _lib.fuzz_extract_one(ctx, ["abc"], [], "WRatio")
    assert idx[0] == -1 and score[0] == 0.0


def test_heavy_row_hand_over(ctx, oracle_mod, monkeypatch):
    """PFZ_K7_HAND_BATCHES=1: every from-string that scores a second batch leaves the rest of its to-groups to the
    continuation launch (16 units starting from the score reached) -- same answers, bit for bit"""
    from polyfuzz_amd import _lib, datasets
    fl, tl = datasets.c3_lists(3000)
    fl = fl[:150]
    monkeypatch.setenv("PFZ_K7_HAND_BATCHES", "1")
    monkeypatch.setenv("PFZ_K7_PARTS", "1")          # (few from-strings would be split over several units: no hand-over there)
    t_dev = _lib.DeviceStrings.upload(ctx, tl)
    for mode in ("WRatio", "partial_ratio", "token_ratio"):
        idx, score = _lib.fuzz_extract_one(ctx, fl, t_dev, mode)
        e_idx, e_score = oracle_mod.fuzz_extract_one(fl, tl, mode)
        np.testing.assert_array_equal(score, e_score, err_msg=mode)
        np.testing.assert_array_equal(idx, e_idx, err_msg=mode)


def test_prepared_lists_equal_the_python_statement(ctx):
    """the device-built forms / plan drive the same scores as tests/k7_prep.py's plain-Python statement drives through the
    CPU build of k7_core.h (tests/test_k7_core_cpu.py): here the end-to-end check is against the oracle on lists with
    tabs, repeated tokens, non-Latin-1 characters (UTF-32 code units), a to-list without any space"""
    from oracle import fuzz_scorers as f
    from polyfuzz_amd import _lib
    fl = ["a\tb  a", "été 　x", "naïve café", "x y z", "  lead and trail  ", "ß", "日本 語", "日本語 日本", "a b a b c"]
    tl = ["b a", "x　été", "cafe naive", "z y x", "trail lead and", "日本", "語 日本", "c b a", "ss"]
    for mode in MODES:
        idx, score = _lib.fuzz_extract_one(ctx, fl, tl, mode)
        e_idx, e_score = f.extract_one_all(fl, tl, f.SCORERS[mode])
        np.testing.assert_array_equal(score, np.array(e_score), err_msg=mode)
        np.testing.assert_array_equal(idx, np.array(e_idx, np.int32), err_msg=mode)
    nospace = ["abc", "abd", "xyz"]                      # the joining space is a symbol even when the to-list has none
    idx, score = _lib.fuzz_extract_one(ctx, ["abc xyz", "abd"], nospace, "WRatio")
    e_idx, e_score = f.extract_one_all(["abc xyz", "abd"], nospace, f.WRatio)
    assert idx.tolist() == e_idx and score.tolist() == e_score


def test_full_size_rows_vs_the_c_oracle(ctx, oracle_mod):
    """random from-rows of the real 20 000 x 20 000 IMDB title lists (config 3's lists, the RapidFuzz default scorer and two
    others), the whole to-list as choices: first best choice and float64 score, bit for bit"""
    import concurrent.futures as cf
    from polyfuzz_amd import _lib, datasets
    fl, tl = datasets.c3_lists()
    f_dev, t_dev = _lib.DeviceStrings.upload(ctx, fl), _lib.DeviceStrings.upload(ctx, tl)
    rows = np.sort(np.random.default_rng(3).choice(len(fl), 96, replace=False))
    sample = [fl[i] for i in rows]
    for mode in ("WRatio", "partial_ratio", "token_set_ratio"):
        idx, score = _lib.fuzz_extract_one(ctx, f_dev, t_dev, mode)
        chunks = np.array_split(np.arange(len(sample)), 32)
        with cf.ThreadPoolExecutor(32) as ex:
            parts = list(ex.map(lambda c: oracle_mod.fuzz_extract_one([sample[i] for i in c], tl, mode), chunks))
        e_idx = np.concatenate([p[0] for p in parts])
        e_score = np.concatenate([p[1] for p in parts])
        np.testing.assert_array_equal(score[rows], e_score, err_msg=mode)
        np.testing.assert_array_equal(idx[rows], e_idx, err_msg=mode)


def test_names_self_match_rows_vs_the_c_oracle(ctx, oracle_mod):
    """RapidFuzz().match(names): all 100 000 company names against themselves under WRatio (every word class at once, the
    long-string classes on the side stream, heavy rows handed over) -- two runs identical, and 64 random rows equal to the
    oracle's answer over the whole list (own first occurrence skipped), bit for bit"""
    import concurrent.futures as cf
    from polyfuzz_amd import _lib, datasets
    names = datasets.load_company_names()
    first = {}
    for j, s in enumerate(names):
        first.setdefault(s, j)
    skip = np.fromiter((first[s] for s in names), np.int32, len(names))
    dev = _lib.DeviceStrings.upload(ctx, names)
    idx, score = _lib.fuzz_extract_one(ctx, dev, dev, "WRatio", skip)
    idx2, score2 = _lib.fuzz_extract_one(ctx, dev, dev, "WRatio", skip)
    np.testing.assert_array_equal(idx, idx2)
    np.testing.assert_array_equal(score, score2)
    rng = np.random.default_rng(8)
    long_rows = np.array([i for i, s in enumerate(names) if len(s) > 64][:8], np.int64)
    rows = np.unique(np.concatenate([rng.choice(len(names), 56, replace=False), long_rows]))

    def one(i):
        return oracle_mod.fuzz_extract_one(names, names, "WRatio", skip=skip, rows=(int(i), int(i) + 1))
    with cf.ThreadPoolExecutor(32) as ex:
        parts = list(ex.map(one, rows))
    e_idx = np.concatenate([p[0] for p in parts])
    e_score = np.concatenate([p[1] for p in parts])
    np.testing.assert_array_equal(score[rows], e_score)
    np.testing.assert_array_equal(idx[rows], e_idx)


def test_four_word_strings_and_editdistance_scorers(ctx):
    """From-strings of 129 .. 256 characters (four 64-bit words: the longest company names have 145) against the
    oracle; EditDistance(scorer=...) takes the same scorers (the reference maps any scorer over all pairs,
    _distance.py:89-102) and keeps first arg-max / un-normalised score."""
    from oracle import fuzz_scorers as f
    from polyfuzz_amd import _lib
    from polyfuzz_amd.models import EditDistance
    rng = np.random.default_rng(2)
    words = ["consolidated", "international", "holdings", "of", "the", "pacific", "northwest", "and", "partners", "llc",
             "limited", "liability", "company", "trust", "fund", "series", "a", "b", "investment", "management"]
    fl = [" ".join(rng.choice(words, size=int(k))) for k in (14, 17, 20, 22, 3, 1)]
    tl = [" ".join(rng.choice(words, size=int(rng.integers(1, 24)))) for _ in range(80)] + [fl[1][:200], fl[2][40:]]
    fl = [s[:256] for s in fl]
    assert max(map(len, fl)) > 128
    for mode in ("WRatio", "partial_ratio", "token_set_ratio"):
        idx, score = _lib.fuzz_extract_one(ctx, fl, tl, mode)
        e_idx, e_score = f.extract_one_all(fl, tl, f.SCORERS[mode])
        np.testing.assert_array_equal(score, np.array(e_score))
        np.testing.assert_array_equal(idx, np.array(e_idx, np.int32))
    df = EditDistance(scorer="token_set_ratio", normalize=False).match(fl, tl)
    e_idx, e_score = f.extract_one_all(fl, tl, f.token_set_ratio)
    assert df["To"].tolist() == [tl[j] for j in e_idx] and df["Similarity"].tolist() == e_score


def test_rapidfuzz_matcher_default_scorer(ctx):
    """RapidFuzz() -- what PolyFuzz("EditDistance") constructs (polyfuzz.py:128-130) -- scores with fuzz.WRatio."""
    from oracle import fuzz_scorers as f
    from polyfuzz_amd.models import RapidFuzz
    fl = ["apple", "apples", "appl", "recal", "house", "similarity"]
    tl = ["apple", "apples", "mouse"]
    for cutoff in (0, 0.9):
        df = RapidFuzz(score_cutoff=cutoff).match(fl, tl)
        e_idx, e_score = f.extract_one_all(fl, tl, f.WRatio)
        assert df["To"].tolist() == [tl[j] if s >= cutoff * 100 else None for j, s in zip(e_idx, e_score)]
        assert df["Similarity"].tolist() == [s / 100 if s >= cutoff * 100 else 0.0 for s in e_score]
    df = RapidFuzz(scorer="partial_ratio").match(fl)
    first = {s: j for j, s in reversed(list(enumerate(fl)))}
    e_idx, e_score = f.extract_one_all(fl, fl, f.partial_ratio, [first[s] for s in fl])
    assert df["To"].tolist() == [fl[j] for j in e_idx] and df["Similarity"].tolist() == [s / 100 for s in e_score]
    with pytest.raises(NotImplementedError):
        RapidFuzz(scorer=lambda a, b: 1.0)


def test_precomputed_oracle_cases(ctx):
    """tests/golden/fuzz_cases.json: 8 seeded list pairs (60 x 150 strings, from-strings up to 128 characters, few and
    many / repeated tokens, small and large alphabets) with the oracle's answers for all seven scorers, made by
    `python tools/k7_stress.py make tests/golden/fuzz_cases.json 8` (five minutes of Python on a CPU, which is
    why it is a fixture): first best choice and float64 score, bit for bit."""
    import json
    import os
    from polyfuzz_amd import _lib
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_cases.json")) as fh:
        cases = json.load(fh)
    assert len(cases) == 8
    for case in cases:
        for mode in MODES:
            idx, score = _lib.fuzz_extract_one(ctx, case["from"], case["to"], mode)
            e_idx, e_score = case["expect"][mode]
            np.testing.assert_array_equal(score, np.array(e_score))
            np.testing.assert_array_equal(idx, np.array(e_idx, np.int32))


def test_reference_rapidfuzz_test_scenarios(ctx):
    """The reference's own tests for this matcher (tests/models/test_rapidfuzz.py:9-36) on the README lists: default
    construction, an explicit scorer, a 0.95 cut-off -- shape, columns and the means they assert."""
    from polyfuzz_amd.models import RapidFuzz
    from_list = ["apple", "apples", "appl", "recal", "house", "similarity"]
    to_list = ["apple", "apples", "mouse"]
    import pandas as pd
    for kwargs, check in (({}, lambda m: m > 0.0), ({"scorer": "ratio"}, lambda m: m > 0.0), ({"score_cutoff": 0.95}, lambda m: m < 0.5)):
        model = RapidFuzz(**kwargs)
        matches = model.match(from_list, to_list)
        assert model.type == "EditDistance" and isinstance(matches, pd.DataFrame) and len(matches) == 6
        assert list(matches.columns) == ["From", "To", "Similarity"]
        assert check(matches.Similarity.mean())


@pytest.mark.parametrize("n_chars", [230, 600])
def test_alphabet_size_and_the_scratch_columns(ctx, oracle_mod, n_chars):
    """The window sweeps stage a to-form in an LDS column of BYTES when the to-list's alphabet has at most 255 symbols, of
    16-bit ranks beyond: lists over 230 and over 600 distinct characters (CJK range, plus the ASCII words), every scorer
    that sweeps windows and the token scorers, bit for bit -- with from-strings much shorter and much longer than the
    to-strings, so that both families of windows run."""
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(n_chars)
    glyphs = [chr(0x4E00 + k) for k in range(n_chars)]

    def mk(n, lo, hi):
        out = []
        for _ in range(n):
            words = []
            for _ in range(int(rng.integers(1, 5))):
                words.append("".join(rng.choice(glyphs, size=int(rng.integers(lo, hi)))))
            out.append(" ".join(words))
        return out
    tl = mk(400, 2, 9) + mk(60, 1, 3)
    # from-strings built from pieces of to-strings (so that scores are high and windows matter), shorter and longer
    fl = []
    for _ in range(60):
        t = tl[int(rng.integers(0, len(tl)))]
        a, b = sorted(int(x) for x in rng.integers(0, len(t) + 1, size=2))
        piece = t[a:b] or t[:3]
        fl.append(piece if rng.random() < 0.5 else piece + " " + tl[int(rng.integers(0, len(tl)))] + " " + "".join(rng.choice(glyphs, size=6)))
    fl += mk(20, 2, 9) + ["", glyphs[0]]
    info_symbols = len(set("".join(tl)))
    assert (info_symbols <= 255) == (n_chars == 230), info_symbols
    for mode in ("WRatio", "partial_ratio", "partial_token_ratio", "token_ratio"):
        idx, score = _lib.fuzz_extract_one(ctx, fl, tl, mode)
        e_idx, e_score = oracle_mod.fuzz_extract_one(fl, tl, mode)
        np.testing.assert_array_equal(score, e_score, err_msg=mode)
        np.testing.assert_array_equal(idx, e_idx, err_msg=mode)


def test_re_train_false_scores_against_the_list_it_is_given(ctx):
    """`match(..., re_train=False)` re-uses the resident to-list only when the list handed in IS that list;
    the reference's RapidFuzz / EditDistance always score against the to_list they are given (PolyFuzz.transform hands them
    `self.to_list`, polyfuzz.py:234-240).  match(X, B) -> a self-match of A -> transform-style match(E, A): against A."""
    from polyfuzz_amd.models import EditDistance, RapidFuzz
    a, b = _lists(11, 40, 60)
    x, e = _lists(12, 30, 25)
    for make in (lambda: RapidFuzz(), lambda: RapidFuzz(scorer="ratio"), lambda: EditDistance(normalize=False)):
        fresh = make().match(e, a)
        m = make()
        m.match(x, b)
        m.match(a)                                            # a fit on one list: self-match, leaves nothing of B behind
        assert m._to_dev is None and m._to_names is None
        assert m.match(e, a, re_train=False).equals(fresh)
        assert m._to_names == tuple(a)                        # (a snapshot of the contents, not the caller's object: ADVICE r4)
        held = m._to_dev
        assert m.match(x, a, re_train=False).equals(make().match(x, a)) and m._to_dev is held      # same object: resident copy
        assert m.match(x, list(a), re_train=False).equals(make().match(x, a)) and m._to_dev is held  # an equal list: too
        assert m.match(x, b, re_train=False).equals(make().match(x, b)) and m._to_dev is not held  # another list: uploaded
        # the caller changes its list IN PLACE and hands the same object over again: the resident copy is stale
        c = list(a)
        m.match(x, c, re_train=False)
        held = m._to_dev
        c[3] = x[0]
        assert m.match(x, c, re_train=False).equals(make().match(x, c)) and m._to_dev is not held
        import numpy as np_
        assert m.match(x, np_.array(c, dtype=object), re_train=False).equals(make().match(x, c))   # an ndarray to-list: no ambiguous `==`
    # a matcher pickled before `reference_self_match` existed
    m = RapidFuzz()
    state = m.__getstate__()
    state.pop("reference_self_match")
    m2 = RapidFuzz.__new__(RapidFuzz)
    m2.__setstate__(state)
    assert m2.reference_self_match is False and m2.match(e, a).equals(RapidFuzz().match(e, a))


def test_rapidfuzz_pin_fixture(ctx):
    """tests/golden/rapidfuzz_pin.json (tests/golden/pin_rapidfuzz.py: boundary cases -- WRatio's gates at exactly 1.5 and
    exactly 8, whitespace-only strings, exotic separators, long needles -- and real titles): `process.extractOne` of every
    a[i] against all of b under all ten rapidfuzz.fuzz scorers, and the score of every pair (a[i], b[i]) as a one-choice
    extractOne.  The file's "source" says whether its numbers are rapidfuzz's own or the oracle's restatement."""
    import json
    import os
    from polyfuzz_amd.models._rapidfuzz import best_choice
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rapidfuzz_pin.json"), encoding="utf-8") as fh:
        pin = json.load(fh)
    a, b = pin["a"], pin["b"]
    assert len(pin["scorers"]) == 10 and len(pin["wratio_ratio8"]["pairs"]) >= 8 and pin["wratio_ratio8"]["chosen"] == "lt8"
    for name in pin["scorers"]:
        idx, score = best_choice(ctx, name, a, b, None, False)
        exp = np.array(pin["extract_one"][name])
        np.testing.assert_array_equal(score, exp[:, 1], err_msg=name)
        np.testing.assert_array_equal(idx, exp[:, 0].astype(np.int32), err_msg=name)
    for name in ("WRatio", "partial_ratio", "token_set_ratio", "partial_token_ratio", "QRatio", "token_sort_ratio"):
        got = [float(best_choice(ctx, name, [x], [y], None, False)[1][0]) for x, y in zip(a, b)]
        np.testing.assert_array_equal(np.array(got), np.array(pin["pair_scores"][name]), err_msg=name)
    for rec in pin["wratio_ratio8"]["pairs"]:                # the chosen reading of the 8x gate, by name
        assert float(best_choice(ctx, "WRatio", [rec["a"]], [rec["b"]], None, False)[1][0]) == rec[pin["wratio_ratio8"]["chosen"]]


def test_rapidfuzz_reference_self_match(ctx):
    """RapidFuzz.match(names, reference_self_match=True): the reference's shared, shrinking list (_rapidfuzz.py:103-104, n_jobs
    = 1) on the device -- skip code -2 - i leaves out every choice up to row i itself.  Against frames of the REFERENCE CLASS
    (tests/golden/rapidfuzz_selfmatch_golden.json: the oracle's scorers stubbed in for rapidfuzz) and, on 1 500 real titles with
    repeats, against the literal restatement over the C scorers; K4 (ratio) and K7 (WRatio, token_set_ratio, partial_ratio)."""
    import json
    import os
    from oracle import reference_path
    from polyfuzz_amd.models import RapidFuzz
    here = os.path.dirname(os.path.abspath(__file__))
    g = json.load(open(os.path.join(here, "golden", "rapidfuzz_selfmatch_golden.json"), encoding="utf-8"))
    for case in g["cases"]:
        names = g["names"][:case["n"]] if "n" in case else g["names"]
        m = RapidFuzz(score_cutoff=case["score_cutoff"], scorer=case["scorer"])
        m.reference_self_match = True
        df = m.match(list(names))
        assert df["From"].tolist() == case["From"]
        assert df["To"].tolist() == case["To"], (case["scorer"], case["score_cutoff"])
        assert df["Similarity"].tolist() == case["Similarity"]
        if "n" not in case:     # the default stays what it was: only the row's own first occurrence is left out
            assert RapidFuzz(score_cutoff=case["score_cutoff"], scorer=case["scorer"]).match(list(names))["To"].tolist() != case["To"]
    from polyfuzz_amd import datasets
    titles = list(datasets.load_movie_titles().values())[0]
    names = titles[:1450] + titles[100:150]            # repeats: list.remove takes the first equal element left
    for scorer, cutoff in (("WRatio", 0.0), ("ratio", 0.55)):
        frm, to, sim = reference_path.rapidfuzz_shared_list_self_match(names, scorer, cutoff)
        df = RapidFuzz(score_cutoff=cutoff, scorer=scorer).match(list(names), reference_self_match=True)
        assert df["To"].tolist() == to
        assert df["Similarity"].tolist() == sim


def test_partial_ratio_windows_vs_third_party_lcs_on_the_device(ctx):
    """K7's window sweep against tests/golden/windows_golden.json directly -- `textdistance.lcsseq` on every window of 388
    (needle, haystack) pairs, no oracle in between: partial_ratio(needle, haystack) must be the maximum over the library's
    window LCS values of the published formula (needle strictly shorter: with equal lengths the roles are also swapped)."""
    from oracle import fuzz_scorers as f
    from polyfuzz_amd import _lib
    from tests.test_fuzz_oracle_cpu import _window_pin
    pairs, windows = _window_pin()
    checked = 0
    for a, b, lcs in pairs:
        if len(a) >= len(b):
            continue
        want = max(f._ratio_of(len(a) + len(w) - 2 * k, len(a) + len(w)) for w, k in zip(windows(a, b), lcs))
        for frm, to in (([a], [b]), ([b], [a])):                      # either string may be the from-string
            idx, score = _lib.fuzz_extract_one(ctx, frm, to, "partial_ratio")
            assert idx[0] == 0 and score[0] == want, (a, b, score[0], want)
        checked += 1
    assert checked > 300
