"""Randomised parity of K7 against oracle/fuzz_scorers.py beyond the test suite's cases: lengths around the 64-bit
word boundary, many / repeated / no tokens, equal lengths (both partial_ratio directions), all seven scorers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polyfuzz_amd
from polyfuzz_amd import _lib
from oracle import fuzz_scorers as f
ctx = polyfuzz_amd.Context.default()
modes = list(_lib.FUZZ_SCORERS)
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
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
    fl, tl = mk(40, 10 if seed % 2 else 14), mk(90, 12)
    fl = [s for s in fl if len(set(s.split())) <= 32]
    for mode in modes:
        idx, score = _lib.fuzz_extract_one(ctx, fl, tl, mode)
        e_idx, e_score = f.extract_one_all(fl, tl, f.SCORERS[mode])
        ok = np.array_equal(score, np.array(e_score)) and np.array_equal(idx, np.array(e_idx, np.int32))
        if not ok:
            bad += 1
            w = np.nonzero((score != np.array(e_score)) | (idx != np.array(e_idx)))[0][:3]
            print("MISMATCH seed", seed, mode, [(fl[i], tl[idx[i]], score[i], tl[e_idx[i]], e_score[i]) for i in w])
    print("seed", seed, "max from-length", max(map(len, fl)), "done")
print("mismatching (seed, mode) combinations:", bad)
