"""
The reference RapidFuzz matcher's SELF-MATCH, run in the build container: polyfuzz/models/_rapidfuzz.py:86-113 executes as
it is written -- `to_list = from_list.copy()`, one shared list, `to_list.remove(from_string)` before every
`process.extractOne` -- so with n_jobs = 1 the list shrinks as the rows are processed and row i is scored against the
strings after it only (the last row against nothing: `None`, 0.0).  polyfuzz_amd.models.RapidFuzz reproduces that on the
device when asked to (`reference_self_match`); this script writes the frames it is held to,
tests/golden/rapidfuzz_selfmatch_golden.json.

rapidfuzz itself is not installable here (no wheel, no network): the module the reference imports is a stub whose
`fuzz.<scorer>` are oracle/fuzz_scorers.py's restatements and whose `process.extractOne` is the documented rule (first
choice with the highest score, `None` below score_cutoff).  What this pins is the reference's OWN list handling -- the part
that is its code --, not the scorers (parity with rapidfuzz stays unpinned).

usage: python tests/golden/make_golden_rapidfuzz_self.py
"""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from oracle import fuzz_scorers  # noqa: E402


def _extract_one(query, choices, scorer=None, processor=None, score_cutoff=None, **kw):
    """rapidfuzz.process.extractOne as documented: (choice, score, index) of the first choice with the highest score, or None
    when no score reaches score_cutoff"""
    best = None
    for j, c in enumerate(choices):
        v = scorer(query, c)
        if score_cutoff is not None and v < score_cutoff:
            continue
        if best is None or v > best[1]:
            best = (c, v, j)
    return best


def _install_stubs():
    sys.modules["seaborn"] = types.ModuleType("seaborn")
    rf = types.ModuleType("rapidfuzz")
    fuzz = types.ModuleType("rapidfuzz.fuzz")
    process = types.ModuleType("rapidfuzz.process")
    for name, f in fuzz_scorers.SCORERS.items():
        setattr(fuzz, name, f)
    process.extractOne = _extract_one
    rf.fuzz, rf.process = fuzz, process
    sys.modules["rapidfuzz"], sys.modules["rapidfuzz.fuzz"], sys.modules["rapidfuzz.process"] = rf, fuzz, process
    return fuzz


def main():
    fuzz = _install_stubs()
    from polyfuzz.models import RapidFuzz          # the reference class itself
    titles = json.load(open(os.path.join(HERE, "titles_self_list.json"), encoding="utf-8"))["from_list"]
    names = titles[:70]
    # repeated strings (list.remove takes the FIRST equal element), an empty string, a one-element tail
    names = names[:20] + [names[3], names[7]] + names[20:50] + ["", names[3]] + names[50:] + [names[0]]
    cases = []
    for scorer, cutoff in (("WRatio", 0.0), ("WRatio", 0.6), ("ratio", 0.0), ("token_set_ratio", 0.5), ("partial_ratio", 0.0)):
        m = RapidFuzz(n_jobs=1, score_cutoff=cutoff, scorer=getattr(fuzz, scorer))
        df = m.match(list(names))
        cases.append({"scorer": scorer, "score_cutoff": cutoff, "From": df["From"].tolist(),
                      "To": [None if t is None else t for t in df["To"].tolist()], "Similarity": [float(x) for x in df["Similarity"]]})
    for n in (1, 2):                              # the shortest lists: the last row has nothing left to match
        m = RapidFuzz(n_jobs=1, scorer=fuzz.WRatio)
        df = m.match(list(names[:n]))
        cases.append({"scorer": "WRatio", "score_cutoff": 0.0, "From": df["From"].tolist(), "To": df["To"].tolist(),
                      "Similarity": [float(x) for x in df["Similarity"]], "n": n})
    out = {"what": "frames of the reference's RapidFuzz(n_jobs=1, ...).match(names) -- its shared, shrinking list -- with the "
                   "oracle's scorers stubbed in for rapidfuzz (tests/golden/make_golden_rapidfuzz_self.py)",
           "names": names, "cases": cases}
    with open(os.path.join(HERE, "rapidfuzz_selfmatch_golden.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("wrote", len(cases), "cases,", len(names), "names;", sum(t is None for t in cases[0]["To"]), "rows without a match in case 0")


if __name__ == "__main__":
    main()
