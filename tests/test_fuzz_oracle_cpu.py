"""The oracle's restatement of the rapidfuzz.fuzz scorers (oracle/fuzz_scorers.py) against every value rapidfuzz
publishes for them (README / API documentation) -- PARITY UNPINNED beyond these -- and its own consistency rules."""
import itertools

import numpy as np
import pytest


def test_published_values():
    from oracle import fuzz_scorers as f
    assert f.ratio("this is a test", "this is a test!") == 96.55172413793103
    assert f.partial_ratio("this is a test", "this is a test!") == 100.0
    assert abs(f.ratio("fuzzy wuzzy was a bear", "wuzzy fuzzy was a bear") - 90.9090909090909) < 1e-12
    assert f.token_sort_ratio("fuzzy wuzzy was a bear", "wuzzy fuzzy was a bear") == 100.0
    assert abs(f.token_sort_ratio("fuzzy was a bear", "fuzzy fuzzy was a bear") - 84.21052631578947) < 1e-12
    assert f.token_set_ratio("fuzzy was a bear", "fuzzy fuzzy was a bear") == 100.0
    assert f.WRatio("this is a test", "this is a new test!!!") == 85.5
    assert abs(f.WRatio("this is a word", "THIS IS A WORD") - 21.42857142857143) < 1e-12
    assert f.QRatio("this is a test", "this is a new test!!!") == 80.0 and f.QRatio("", "") == 0.0
    assert f.WRatio("", "x") == 0.0 and f.partial_ratio("", "") == 100.0 and f.partial_ratio("abc", "") == 0.0
    assert f.token_set_ratio("", "a") == 0.0 and f.partial_token_ratio("a b", "   ") == 0.0


def test_relations_between_the_scorers():
    """token_ratio = max(token_sort, token_set); partial_token_ratio = 100 on a common token, else the max of its two
    parts; partial_ratio >= ratio-of-the-best-window by construction, symmetric; WRatio's two branches."""
    from oracle import fuzz_scorers as f
    words = ["new", "york", "mets", "braves", "the", "atlanta", "vs", "a", "bb", "new"]
    strs = [" ".join(c) for n in (1, 2, 3) for c in itertools.islice(itertools.permutations(words, n), 0, 40, 7)] + ["", "x", "mets  new"]
    for a, b in itertools.islice(itertools.product(strs, strs), 0, 600, 5):
        assert f.token_ratio(a, b) == max(f.token_sort_ratio(a, b), f.token_set_ratio(a, b))
        assert f.partial_ratio(a, b) == f.partial_ratio(b, a)
        assert f.partial_ratio(a, b) >= f.ratio(a, b) - 1e-9 or not a or not b
        if set(a.split()) & set(b.split()):
            assert f.partial_token_ratio(a, b) == 100.0 and f.partial_token_set_ratio(a, b) == 100.0
        w = f.WRatio(a, b)
        assert 0.0 <= w <= 100.0 and (w >= f.ratio(a, b) or not a or not b)


def test_extract_one_rules():
    from oracle import fuzz_scorers as f
    idx, score = f.extract_one_all(["apple", "zzz", ""], ["apples", "apple", "apple"], f.ratio)
    assert idx == [1, 0, 0] and score[0] == 100.0 and score[2] == 0.0        # first best; a zero score still picks the first choice
    idx, score = f.extract_one_all(["apple"], ["apple", "apple"], f.ratio, skip=[0])
    assert idx == [1] and f.extract_one_all(["a"], [], f.ratio) == ([-1], [0.0])


def test_bit_vector_lcs_equals_the_dp():
    import numpy as np
    from oracle import fuzz_scorers as f
    rng = np.random.default_rng(0)
    for alphabet in ("ab", "abc ", "abcdefghijklmnopqrstuvwxyz "):
        for _ in range(1500):
            a = "".join(rng.choice(list(alphabet), size=int(rng.integers(0, 90))))
            b = "".join(rng.choice(list(alphabet), size=int(rng.integers(0, 90))))
            assert f.lcs_len(a, b) == f.lcs_len_dp(a, b), (a, b)
    assert f.lcs_len("", "abc") == 0 and f.lcs_len("abc", "abc") == 3


def test_c_restatement_equals_the_python_one_bit_for_bit(oracle_mod):
    """oracle/fuzz_scorers.c (the fast checker of K7 at full list sizes, bench.py's CPU arm) == oracle/fuzz_scorers.py:
    every scorer, real titles and names, empty / blank / repeated-token / non-ASCII-whitespace strings; extractOne with
    and without a skipped choice."""
    import random
    import numpy as np
    from oracle import fuzz_scorers as f
    from polyfuzz_amd import datasets, synth
    fl, tl = datasets.c3_lists(400)
    rng = random.Random(1)
    names = fl + tl + synth.company_names(150, 3) + ["", " ", "a", "a a", "b a a", "  x  y ", "the the", "new york mets",
                                                      "new york mets vs atlanta braves", "été　x", "a\tb\nc\x1fd\x85e\xa0f"]
    for name, fn in f.SCORERS.items():
        for _ in range(400):
            a, b = rng.choice(names), rng.choice(names)
            if rng.random() < 0.25:
                b = a[:rng.randint(0, len(a))] + " " + rng.choice(names)[:5]
            assert fn(a, b) == oracle_mod.fuzz_score(a, b, name), (name, a, b)
    a_list, b_list = fl[:12] + ["", "x"], tl[:150] + ["", "The"]
    for name in ("WRatio", "token_set_ratio", "partial_ratio", "QRatio"):
        idx, score = f.extract_one_all(a_list, b_list, f.SCORERS[name])
        c_idx, c_score = oracle_mod.fuzz_extract_one(a_list, b_list, name)
        assert idx == c_idx.tolist() and score == c_score.tolist(), name
    own = fl[:40] + fl[:5]
    first = {}
    for j, s in enumerate(own):
        first.setdefault(s, j)
    skip = [first[s] for s in own]
    idx, score = f.extract_one_all(own, own, f.WRatio, skip=skip)
    c_idx, c_score = oracle_mod.fuzz_extract_one(own, own, "WRatio", skip=skip, rows=(3, 45))
    assert idx[3:] == c_idx.tolist() and score[3:] == c_score.tolist()
    assert np.array_equal(oracle_mod.fuzz_extract_one(["a"], [], "WRatio")[0], [-1])


def _pin():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rapidfuzz_pin.json"), encoding="utf-8") as fh:
        return json.load(fh)


def test_rapidfuzz_pin_fixture_vs_both_restatements(oracle_mod):
    """tests/golden/rapidfuzz_pin.json (made by tests/golden/pin_rapidfuzz.py; "source" = rapidfuzz's own numbers once the
    script has run where the library is importable, the oracle's until then): every pair score and every extractOne under
    all ten scorers, against oracle/fuzz_scorers.py AND oracle/fuzz_scorers.c; the boundary cases are there by construction
    (length ratio exactly 1.5 -> the partial branch, exactly 8 -> recorded under both readings, the chosen one named)."""
    from oracle import fuzz_scorers as f
    pin = _pin()
    a, b = pin["a"], pin["b"]
    assert sum(1 for x, y in zip(a, b) if x and y and 2 * max(len(x), len(y)) == 3 * min(len(x), len(y))) >= 7
    assert sum(1 for x, y in zip(a, b) if x and y and max(len(x), len(y)) == 8 * min(len(x), len(y))) >= 9
    for name in pin["scorers"]:
        for i, (x, y) in enumerate(zip(a, b)):
            assert f.SCORERS[name](x, y) == pin["pair_scores"][name][i], (name, x, y)
            assert oracle_mod.fuzz_score(x, y, name) == pin["pair_scores"][name][i], (name, x, y)
        idx, score = oracle_mod.fuzz_extract_one(a, b, name)
        exp = np.array(pin["extract_one"][name])
        np.testing.assert_array_equal(score, exp[:, 1])
        np.testing.assert_array_equal(idx, exp[:, 0].astype(np.int32))
    r8 = pin["wratio_ratio8"]
    assert r8["chosen"] == "lt8"
    for rec in r8["pairs"]:
        assert f.WRatio(rec["a"], rec["b"]) == rec["lt8"] and rec["le8"] >= rec["lt8"]
        assert rec["le8"] != rec["lt8"] or f.ratio(rec["a"], rec["b"]) == rec["lt8"]     # the gate matters on these pairs


def test_real_rapidfuzz_when_importable(oracle_mod):
    """Where rapidfuzz IS importable this is the pin itself: the library against the oracle on the fixture's strings."""
    pytest.importorskip("rapidfuzz")
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import pin_rapidfuzz
    a, b = pin_rapidfuzz.strings()
    assert pin_rapidfuzz.diff(pin_rapidfuzz.from_rapidfuzz(a, b), pin_rapidfuzz.from_oracle(a, b)) == []


def test_shared_list_self_match_restatement_equals_the_reference_class():
    """oracle.reference_path.rapidfuzz_shared_list_self_match (the literal restatement of _rapidfuzz.py:86-113 with n_jobs = 1:
    one shared list that shrinks) against frames the REFERENCE CLASS produced in the build container with the oracle's scorers
    stubbed in for rapidfuzz (tests/golden/make_golden_rapidfuzz_self.py): From / To cell for cell, Similarity bit for bit --
    with the Python scorers and with the C ones."""
    import json
    import os
    from oracle import reference_path
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rapidfuzz_selfmatch_golden.json"),
                       encoding="utf-8"))
    for case in g["cases"]:
        names = g["names"][:case["n"]] if "n" in case else g["names"]
        for use_c in (True, False):
            if not use_c and case["scorer"] not in ("ratio", "WRatio") :
                continue                                  # (the Python scorers are slow: two scorers do)
            frm, to, sim = reference_path.rapidfuzz_shared_list_self_match(names, case["scorer"], case["score_cutoff"], use_c=use_c)
            assert frm == case["From"]
            assert to == case["To"], (case["scorer"], case["score_cutoff"])
            assert sim == case["Similarity"]
        assert to[-1] is None and sim[-1] == 0.0          # the last row has nothing left to match


def _window_pin():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "windows_golden.json")
    fix = json.load(open(path, encoding="utf-8"))
    assert fix["textdistance"] and len(fix["pairs"]) > 300

    def windows(s1, s2):          # the order the published algorithm walks them (fuzz_py._partial_ratio_impl)
        m, n = len(s1), len(s2)
        return [s2[:k] for k in range(1, m)] + [s2[i:i + m] for i in range(0, n - m + 1)] + [s2[i:] for i in range(n - m + 1, n)]
    return fix["pairs"], windows


def test_partial_ratio_windows_vs_third_party_lcs(oracle_mod):
    """the integer core of partial_ratio's window sweep held to a LIBRARY, window by window --
    tests/golden/windows_golden.json is `textdistance.lcsseq` on every window of 388 (needle, haystack) pairs (13 859
    windows; made by tests/golden/make_golden_windows.py under the container's conda interpreter).  Both restatements must
    give every window's LCS and, from those integers by the one published formula, the pair's score."""
    from oracle import fuzz_scorers as f
    pairs, windows = _window_pin()
    n_win = 0
    for a, b, lcs in pairs:
        ws = windows(a, b)
        assert len(ws) == len(lcs)
        for w, k in zip(ws, lcs):
            assert f.lcs_len(a, w) == k and f.lcs_len_dp(a, w) == k, (a, w, k)
            assert oracle_mod.fuzz_score(a, w, "ratio") == f._ratio_of(len(a) + len(w) - 2 * k, len(a) + len(w)), (a, w)      # (oracle/fuzz_scorers.c)
        n_win += len(ws)
        want = max([f._ratio_of(len(a) + len(w) - 2 * k, len(a) + len(w)) for w, k in zip(ws, lcs)] + [0.0])
        assert f._partial_ratio_impl(a, b) == want, (a, b)
        if len(a) < len(b):          # (equal lengths: partial_ratio also runs with the roles swapped)
            assert f.partial_ratio(a, b) == want and oracle_mod.fuzz_score(a, b, "partial_ratio") == want, (a, b)
    assert n_win > 13000
