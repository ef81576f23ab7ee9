"""Randomised parity of K7 against oracle/fuzz_scorers.py beyond the test suite's cases: lengths around the 64-bit
word boundary, many / repeated / no tokens, equal lengths (both partial_ratio directions), all seven scorers.

Two steps, because the oracle is slow Python and GPU minutes are not for it:
  python tools/k7_stress.py make <file> [seeds]     (anywhere)  cases + the oracle's answers -> JSON
  python tools/k7_stress.py check <file>            (GPU box)   K7 against them
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

MODES = ["WRatio", "partial_ratio", "token_set_ratio", "token_ratio", "partial_token_sort_ratio",
         "partial_token_set_ratio", "partial_token_ratio"]


def make(path, seeds):
    from oracle import fuzz_scorers as f
    cases = []
    for seed in range(seeds):
        rng = np.random.default_rng(100 + seed)
        alpha = list("abcde") if seed % 2 else list("abcdefghijklmnopqrstuvwxyz")

        def word():
            return "".join(rng.choice(alpha, size=int(rng.integers(1, 12 if seed % 3 else 30))))
        vocab = [word() for _ in range(12)]

        def mk(n, kmax):
            out = []
            for _ in range(n):
                k = int(rng.integers(0, kmax))
                s = " ".join(rng.choice(vocab, size=k)) if k else ""
                if rng.random() < 0.3:
                    s = s + " " + word()
                out.append(s[:128].strip() if len(s) > 128 else s)
            return out
        fl, tl = mk(60, 10 if seed % 2 else 14), mk(150, 12)
        fl = [s for s in fl if len(set(s.split())) <= 32]
        exp = {m: f.extract_one_all(fl, tl, f.SCORERS[m]) for m in MODES}
        cases.append({"from": fl, "to": tl, "expect": exp})
        print("seed", seed, "max from-length", max(map(len, fl)), flush=True)
    json.dump(cases, open(path, "w"))


def check(path):
    import polyfuzz_amd
    from polyfuzz_amd import _lib
    ctx = polyfuzz_amd.Context.default()
    bad = 0
    for ci, case in enumerate(json.load(open(path))):
        fl, tl = case["from"], case["to"]
        for m in MODES:
            idx, score = _lib.fuzz_extract_one(ctx, fl, tl, m)
            e_idx, e_score = case["expect"][m]
            if not (np.array_equal(score, np.array(e_score)) and np.array_equal(idx, np.array(e_idx, np.int32))):
                bad += 1
                w = np.nonzero((score != np.array(e_score)) | (idx != np.array(e_idx)))[0][:3]
                print("MISMATCH case", ci, m, [(fl[i], tl[idx[i]], score[i], tl[e_idx[i]], e_score[i]) for i in w])
    print("cases", ci + 1, "x", len(MODES), "scorers; mismatching combinations:", bad)


if __name__ == "__main__":
    if sys.argv[1] == "make":
        make(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 6)
    else:
        check(sys.argv[2])
