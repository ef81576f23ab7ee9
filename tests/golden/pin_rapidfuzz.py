"""Pin script for the rapidfuzz family (reference call sites: polyfuzz/models/_distance.py:4,32,98 `fuzz.ratio`;
_rapidfuzz.py:3,48,106-108 `process.extractOne(scorer=fuzz.WRatio, score_cutoff=...)`).

rapidfuzz (setup.py:20, `rapidfuzz>=0.13.1`) is NOT installable in the build container (no wheel, no network), so
oracle/fuzz_scorers.{py,c} restate its published semantics and say "parity unpinned".  This script is what closes that
the moment the library is at hand -- on any machine:

    python tests/golden/pin_rapidfuzz.py            # rapidfuzz importable: regenerate from the REAL library, diff vs oracle
    python tests/golden/pin_rapidfuzz.py --oracle   # not importable: (re)write the fixture from the oracle ("source": "oracle")
    python tests/golden/pin_rapidfuzz.py --check    # diff only, write nothing; exit code 1 on any difference

It writes tests/golden/rapidfuzz_pin.json: the strings below (boundary cases first, then real titles), per scorer the score
of every pair (a[i], b[i]) and `process.extractOne` of every a[i] against ALL of b (first best index + score).
tests/test_fuzz_oracle_cpu.py holds oracle/fuzz_scorers.{py,c} to the file, tests/test_fuzz_gpu.py holds K4 / K7 to it; its
"source" field says whether the numbers are the library's or the restatement's.  With rapidfuzz importable,
tests/test_fuzz_oracle_cpu.py::test_real_rapidfuzz_when_importable also runs this comparison live.

The boundary cases: WRatio's length-ratio gates at EXACTLY 1.5 and EXACTLY 8, just below and just
above both; empty and whitespace-only strings; tokens separated by tabs / NBSP / U+2003 / U+001C (Python's str.split set);
repeated tokens; needles longer than 64 characters (rapidfuzz's long-needle partial_ratio); equal-length partial_ratio.
The 8x gate is the one place where the published sources disagree: rapidfuzz-cpp's `fuzz_impl.hpp` (what the installed C++
extension runs) reads `PARTIAL_SCALE = len_ratio < 8.0 ? 0.9 : 0.6`, while thefuzz's `fuzz.WRatio` and rapidfuzz's
pure-Python fallback `fuzz_py.py` read `len_ratio <= 8` -> 0.9 (thefuzz: `elif len_ratio > 8: partial_scale = .6`).  The oracle
and the kernel follow the C++ extension (`< 8.0`): it is what `from rapidfuzz import fuzz` resolves to wherever a wheel
installs.  "wratio_ratio8" records those pairs under BOTH readings and names the chosen one.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

SCORERS = ("ratio", "QRatio", "partial_ratio", "token_sort_ratio", "token_set_ratio", "token_ratio",
           "partial_token_sort_ratio", "partial_token_set_ratio", "partial_token_ratio", "WRatio")
CHOSEN_8X = "lt8"     # rapidfuzz-cpp fuzz_impl.hpp: (len_ratio < 8.0) ? 0.9 : 0.6


def strings():
    """(a, b): two equally long lists; pair i is (a[i], b[i]), extractOne runs a[i] against all of b."""
    rnd = random.Random(8)
    def pad(s, n, ch):                      # s padded on the right with `ch` to n characters
        assert len(s) <= n
        return s + ch * (n - len(s))
    exact15 = [("abcdef", "abcd"), ("abcd", "abcdef"), ("abc", "ab"), ("a" * 12, "a" * 8),
               ("new york mets vs atlanta", "york new atlanta"), ("ab cd ef gh ijk", "ef ab cd q"),
               ("the mets of queens", "mets the of ")]
    below15 = [("abcdefghij", "abcdefg"), ("abcdefg", "abcdefghij"), ("new york mets ny", "mets new york"), ("ab cd ef g", "ef ab cd")]
    above15 = [("abcdefghijk", "abcdefg"), ("new york mets vs nl", "mets new york")]
    exact8 = [("ab", pad("ab", 16, "c")), (pad("ab", 16, "c"), "ab"), ("a", "xxxaxxxx"), ("xy z", pad("the xy z of it all ", 32, "q")),
              ("abc", pad("zzabczz", 24, "y")), ("fuzz", pad("fuzzy wuzzy was a bear ", 32, "k")), ("yank", "new york yankees of the bronx ny "[:32]),
              ("mets", pad("the new york mets of flushing", 32, " ")[:31] + "q"), ("wuzzy fuzzy", pad("fuzzy wuzzy was a bear and had no hair ", 88, "w"))]
    near8 = [("ab", pad("ab", 15, "c")), ("ab", pad("ab", 17, "c")), ("abcd", pad("xx abcd", 31, "y")), ("abcd", pad("xx abcd", 33, "y"))]
    for x, y in exact15:
        assert 2 * max(len(x), len(y)) == 3 * min(len(x), len(y)), (x, y)
    for x, y in exact8:
        assert max(len(x), len(y)) == 8 * min(len(x), len(y)), (x, y)
    pairs = exact15 + below15 + above15 + exact8 + near8 + [
        # empty / whitespace-only / exotic separators
        ("", ""), ("", "abc"), ("abc", ""), ("   ", "abc"), ("\t", "\t"), (" ", "  "), ("a b", "a\tb"), ("a b", "b\u00a0a"),
        ("x y z", "z\u2003y x"), ("p\x1cq", "q p"), ("a  b   c", "c b a"), (" lead", "lead "), ("a\nb", "a b"),
        ("a\u200bb", "a b"),                  # ZERO WIDTH SPACE is NOT whitespace for str.split
        # repeated tokens, subsets, disjoint sets
        ("a a a b", "a b"), ("fuzzy was a bear", "fuzzy fuzzy was a bear"), ("new york mets", "mets york new"),
        ("new york mets vs atlanta braves", "atlanta braves vs new york mets"), ("abc def", "ghi jkl"), ("abc", "abc def"),
        ("this is a test", "this is a test!"), ("this is a test", "this is a new test!!!"), ("this is a word", "THIS IS A WORD"),
        ("fuzzy wuzzy was a bear", "wuzzy fuzzy was a bear"), ("lewenstein", "levenshtein"),
        # equal lengths (partial_ratio scores both directions), long needles (> 64 symbols)
        ("abcdxyz", "xyzabcd"), ("abab", "baba"), ("k" * 70 + "abc", "abc" + "k" * 70), ("ab" * 40, "ba" * 45),
        ("the " * 20 + "end", "start " + "the " * 25), ("x" * 65, "y" + "x" * 130), ("long needle " * 8, "needle long " * 30),
    ]
    d = json.load(open(os.path.join(HERE, "titles_lists.json"), encoding="utf-8"))
    fl, tl = d["from_list"], d["to_list"]
    for a in rnd.sample(fl, 60):                       # real titles (incl. non-ASCII), one near-duplicate each third
        b = rnd.choice(tl)
        if rnd.random() < 0.33:
            toks = a.split()
            rnd.shuffle(toks)
            b = " ".join(toks + [rnd.choice(tl).split()[0]]) if toks else b
        pairs.append((a, b))
    return [p[0] for p in pairs], [p[1] for p in pairs]


def wratio_le8(fz, s1, s2):
    """WRatio as thefuzz / fuzz_py.py read the 8x gate (`len_ratio <= 8` keeps 0.9): the reading NOT chosen."""
    if not s1 or not s2:
        return 0.0
    l1, l2 = len(s1), len(s2)
    lr = l1 / l2 if l1 > l2 else l2 / l1
    end = fz.ratio(s1, s2)
    if lr < 1.5:
        return max(end, fz.token_ratio(s1, s2) * 0.95)
    ps = 0.9 if lr <= 8.0 else 0.6
    end = max(end, fz.partial_ratio(s1, s2) * ps)
    return max(end, fz.partial_token_ratio(s1, s2) * 0.95 * ps)


def from_oracle(a, b):
    from oracle import fuzz_scorers as fz
    out = {"source": "oracle (oracle/fuzz_scorers.py; PARITY UNPINNED until regenerated with rapidfuzz importable)"}
    out["pair_scores"] = {s: [fz.SCORERS[s](x, y) for x, y in zip(a, b)] for s in SCORERS}
    out["extract_one"] = {}
    for s in SCORERS:
        idx, score = fz.extract_one_all(a, b, fz.SCORERS[s])
        out["extract_one"][s] = [[int(i), float(v)] for i, v in zip(idx, score)]
    return out


def from_rapidfuzz(a, b):
    import rapidfuzz
    from rapidfuzz import fuzz, process
    out = {"source": f"rapidfuzz {rapidfuzz.__version__}"}
    out["pair_scores"] = {s: [float(getattr(fuzz, s)(x, y, processor=None)) for x, y in zip(a, b)] for s in SCORERS}
    out["extract_one"] = {}
    for s in SCORERS:
        rows = []
        for x in a:
            r = process.extractOne(x, b, scorer=getattr(fuzz, s), processor=None)
            rows.append([int(r[2]), float(r[1])] if r is not None else [-1, 0.0])
        out["extract_one"][s] = rows
    return out


def ratio8_record(a, b):
    from oracle import fuzz_scorers as fz
    rec = []
    for x, y in zip(a, b):
        if x and y and max(len(x), len(y)) == 8 * min(len(x), len(y)):
            rec.append({"a": x, "b": y, "lt8": fz.WRatio(x, y), "le8": wratio_le8(fz, x, y)})
    return {"chosen": CHOSEN_8X, "why": "rapidfuzz-cpp fuzz_impl.hpp: PARTIAL_SCALE = (len_ratio < 8.0) ? 0.9 : 0.6 -- the "
            "C++ extension is what `from rapidfuzz import fuzz` resolves to; thefuzz and rapidfuzz's fuzz_py.py read <= 8",
            "pairs": rec}


def diff(x, y):
    bad = []
    for s in SCORERS:
        for i, (u, v) in enumerate(zip(x["pair_scores"][s], y["pair_scores"][s])):
            if u != v:
                bad.append(("pair", s, i, u, v))
        for i, (u, v) in enumerate(zip(x["extract_one"][s], y["extract_one"][s])):
            if u[1] != v[1] or (u[0] != v[0]):
                bad.append(("extract_one", s, i, u, v))
    return bad


def main(argv):
    a, b = strings()
    path = os.path.join(HERE, "rapidfuzz_pin.json")
    orc = from_oracle(a, b)
    try:
        import rapidfuzz  # noqa: F401
        have = True
    except ImportError:
        have = False
    if have and "--oracle" not in argv:
        real = from_rapidfuzz(a, b)
        bad = diff(real, orc)
        print(f"{real['source']} vs oracle/fuzz_scorers.py: {len(bad)} differences over {len(a)} pairs x {len(SCORERS)} scorers "
              f"(+ extractOne of every a against all {len(b)} b)")
        for kind, s, i, u, v in bad[:40]:
            print(f"  {kind:11s} {s:26s} #{i}: rapidfuzz {u!r}  oracle {v!r}   a={a[i]!r} b={b[i] if kind == 'pair' else '(all)'!r}")
        rec = real
    elif "--check" in argv:
        cur = json.load(open(path, encoding="utf-8"))
        bad = diff(cur, orc) if cur["a"] == a and cur["b"] == b else [("strings changed",) * 5]
        print(f"fixture ({cur['source']}) vs oracle: {len(bad)} differences")
        rec = None
    else:
        print("rapidfuzz is not importable here: writing the ORACLE's answers (source says so; parity stays unpinned)")
        bad, rec = [], orc
    if rec is not None and "--check" not in argv:
        rec.update({"made_by": "tests/golden/pin_rapidfuzz.py", "scorers": list(SCORERS), "a": a, "b": b,
                    "wratio_ratio8": ratio8_record(a, b)})
        with open(path, "w", encoding="utf-8") as f:
            json.dump(rec, f, ensure_ascii=False, separators=(",", ":"))
        print(f"{len(a)} pairs -> {path} ({os.path.getsize(path)} bytes), {len(rec['wratio_ratio8']['pairs'])} pairs at exactly 8x")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
