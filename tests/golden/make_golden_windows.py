"""Third-party pin for the integer core of the WINDOW SWEEP of `rapidfuzz.fuzz.partial_ratio` (reference call sites:
polyfuzz/models/_rapidfuzz.py:48,106-108 -- WRatio runs it for pairs whose lengths differ by 1.5 x or more).

rapidfuzz itself is not installable in the build container.  What this records is what `textdistance.lcsseq` 4.2.1 says
about the longest common subsequence of the shorter string with EVERY window of the longer one that the published
algorithm looks at (rapidfuzz 3.x, fuzz_py._partial_ratio_impl): the prefixes of the longer string shorter than the
needle, its windows of the needle's length, its suffixes shorter than the needle.  From those integers the score follows by
one formula -- max over the windows of (1 - (|s1| + |w| - 2 lcs) / (|s1| + |w|)) * 100 -- so the oracle's
(oracle/fuzz_scorers.{py,c}) and the kernel's (K7) window arithmetic is held to a library, window by window, on the CPU and
on the GPU.  What stays restated is WHICH windows the library scores (its character-set shortcut only skips windows that
another one dominates) and the float formula.

    /opt/conda/bin/python3.9 tests/golden/make_golden_windows.py      (the interpreter that has textdistance)

-> tests/golden/windows_golden.json: [[needle, haystack, [lcs of every window, in the order prefixes / windows / suffixes]], ...]
"""
import json
import os
import random
import sys

import numpy

if not hasattr(numpy, "int"):
    numpy.int = int
import textdistance    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def windows(s1, s2):
    """the windows of s2 (len(s1) <= len(s2)) in the order the published algorithm walks them"""
    m, n = len(s1), len(s2)
    out = [s2[:k] for k in range(1, m)]                   # prefixes shorter than the needle
    out += [s2[i:i + m] for i in range(0, n - m + 1)]     # full windows
    out += [s2[i:] for i in range(n - m + 1, n)]          # suffixes shorter than the needle
    return out


def pairs():
    rnd = random.Random(20260925)
    d = json.load(open(os.path.join(HERE, "titles_lists.json"), encoding="utf-8"))
    titles = d["from_list"] + d["to_list"]
    out = [("abcd", "xxabcdxx"), ("ab", "ba"), ("a", "a"), ("yank", "new york yankees"), ("mets", "the new york mets of flushing"),
           ("fuzzy wuzzy", "wuzzy fuzzy was a bear"), ("x" * 70, "y" * 3 + "x" * 80), ("ab" * 35, "ba" * 50)]
    short = [t for t in titles if 3 <= len(t) <= 12]
    long_ = [t for t in titles if len(t) >= 18]
    for _ in range(160):                                   # real titles: a short one against a long one (the WRatio case)
        out.append((rnd.choice(short), rnd.choice(long_)))
    for _ in range(60):                                    # a word of the long title, mangled
        b = rnd.choice(long_)
        toks = b.split()
        w = list(rnd.choice(toks))
        if w:
            w[rnd.randrange(len(w))] = rnd.choice("aeiouyz")
        out.append(("".join(w) or "a", b))
    for alpha, n in (("ab", 60), ("abc ", 60), ("abcdefgh", 40)):      # small alphabets: the LCS is far from trivial
        for _ in range(n):
            a = "".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 14)))
            b = "".join(rnd.choice(alpha) for _ in range(rnd.randint(len(a), 60)))
            out.append((a, b))
    return [(a, b) if len(a) <= len(b) else (b, a) for a, b in out]


def main():
    rows = []
    total = 0
    for a, b in pairs():
        lcs = [len(textdistance.lcsseq(a, w)) if a and w else 0 for w in windows(a, b)]
        total += len(lcs)
        rows.append([a, b, lcs])
    path = os.path.join(HERE, "windows_golden.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump({"made_by": "tests/golden/make_golden_windows.py", "python": sys.version.split()[0],
                   "textdistance": textdistance.__version__, "pairs": rows}, f, ensure_ascii=False, separators=(",", ":"))
    print(f"{len(rows)} pairs, {total} windows -> {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
