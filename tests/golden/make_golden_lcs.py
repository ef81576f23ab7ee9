"""Third-party pin for the integer core of `rapidfuzz.fuzz.ratio` (reference call sites: polyfuzz/models/_distance.py:32,98,
_rapidfuzz.py:106-108): the length of the longest common subsequence and the Indel distance |a| + |b| - 2 LCS.

rapidfuzz itself is not installable in the build container (no wheel, no network).  Two INDEPENDENT third-party
implementations are, in the container's conda environment, and this script records what THEY say:
  * `textdistance.lcsseq` 4.2.1 (pure-Python DP with back-tracking; the LCS string's length), and
  * `nltk.edit_distance(a, b, substitution_cost=2)` 3.6.5 (Levenshtein DP; a substitution priced as delete + insert IS the
    Indel distance).
Run it with that interpreter (it is the only one that has the two packages):

    /opt/conda/bin/python3.9 tests/golden/make_golden_lcs.py

It writes tests/golden/lcs_golden.json: [[a, b, lcs_textdistance, indel_nltk], ...].  tests/test_oracle_cpu.py holds
oracle/indel.c, oracle/fuzz_scorers.{py,c} to these numbers; tests/test_indel_gpu.py holds K4 to them.
The pairs: config 3's title fixture (tests/golden/titles_lists.json, incl. non-ASCII titles), near-duplicates of those
titles (edits, token swaps), seeded random strings over small alphabets (where the LCS is far from trivial), the
reference README's lists, empty strings.
"""
import json
import os
import random
import sys

import numpy

if not hasattr(numpy, "int"):          # textdistance 4.2.1 predates numpy 1.24's removal of the alias (script-local shim)
    numpy.int = int
import nltk            # noqa: E402
import textdistance    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def pairs():
    rnd = random.Random(20260924)
    d = json.load(open(os.path.join(HERE, "titles_lists.json"), encoding="utf-8"))
    fl, tl = d["from_list"], d["to_list"]
    out = []
    readme_from = ["apple", "apples", "appl", "recal", "house", "similarity"]
    readme_to = ["apple", "apples", "mouse"]
    out += [(a, b) for a in readme_from for b in readme_to]
    out += [("", ""), ("", "a"), ("abc", ""), ("a", "a"), ("a", "b"), ("ab", "ba"), ("this is a test", "this is a test!"),
            ("lewenstein", "levenshtein"), ("x" * 64, "x" * 65), ("ab" * 40, "ba" * 40), ("a" * 70, "b" * 70)]
    for a in fl:                                           # real titles against real titles
        for b in rnd.sample(tl, 4):
            out.append((a, b))
    for a in rnd.sample(fl + tl, 300):                     # near-duplicates: the pairs a matcher exists for
        s = list(a)
        for _ in range(rnd.randint(1, 4)):
            op = rnd.randint(0, 3)
            p = rnd.randint(0, max(len(s) - 1, 0))
            if op == 0 and s:
                del s[p]
            elif op == 1:
                s.insert(p, rnd.choice("aeiou tnsé"))
            elif op == 2 and s:
                s[p] = rnd.choice("aeiou tnsÅ")
            else:
                toks = "".join(s).split()
                rnd.shuffle(toks)
                s = list(" ".join(toks))
        out.append((a, "".join(s)))
    for alpha, lo, hi, n in (("ab", 0, 40, 500), ("abc", 0, 70, 400), ("abcdefgh ", 1, 130, 400),
                             ("aàáâãäå", 1, 30, 200)):
        for _ in range(n):
            a = "".join(rnd.choice(alpha) for _ in range(rnd.randint(lo, hi)))
            b = "".join(rnd.choice(alpha) for _ in range(rnd.randint(lo, hi)))
            out.append((a, b))
    return out


def main():
    rows = []
    for a, b in pairs():
        lcs = len(textdistance.lcsseq(a, b)) if a and b else 0
        dist = nltk.edit_distance(a, b, substitution_cost=2, transpositions=False)
        # the two libraries are independent of each other; they must agree before either pins anything
        assert dist == len(a) + len(b) - 2 * lcs, (a, b, lcs, dist)
        rows.append([a, b, lcs, dist])
    path = os.path.join(HERE, "lcs_golden.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump({"made_by": "tests/golden/make_golden_lcs.py",
                   "python": sys.version.split()[0],
                   "textdistance": textdistance.__version__, "nltk": nltk.__version__,
                   "pairs": rows}, f, ensure_ascii=False, separators=(",", ":"))
    print(f"{len(rows)} pairs -> {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
