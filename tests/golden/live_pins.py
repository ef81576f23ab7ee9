"""The library pins, ARMED: wherever rapidfuzz / sparse_dot_topn happen to be importable -- the driver's GPU boxes are not the
build container -- `live_pins()` runs the comparisons of pin_rapidfuzz.py / pin_sparse_dot_topn.py against the REAL libraries
on the spot (nothing is written) and reports the number of differences; where they are not importable it says so.  Called by
bench.py (the `library_pins` record of the line) and by __graft_entry__.smoke(): checker code, like the oracle it checks.
Reference call sites: polyfuzz/models/_rapidfuzz.py:3,48,106-108, _distance.py:4,32 (rapidfuzz); _utils.py:9,82 (sparse_dot_topn)."""
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _importable(name):
    try:
        importlib.import_module(name)
        return True
    except Exception:
        return False


def live_pins():
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    out = {"rapidfuzz_importable": _importable("rapidfuzz"), "sparse_dot_topn_importable": _importable("sparse_dot_topn"),
           "rapidfuzz_vs_oracle_differences": None, "sparse_dot_topn_rows_differing": None}
    if out["rapidfuzz_importable"]:
        try:
            pin = importlib.import_module("pin_rapidfuzz")
            a, b = pin.strings()
            bad = pin.diff(pin.from_rapidfuzz(a, b), pin.from_oracle(a, b))
            out["rapidfuzz_vs_oracle_differences"] = len(bad)
            out["rapidfuzz_checked"] = f"{len(a)} pairs x {len(pin.SCORERS)} scorers + extractOne of every a against all b"
            out["rapidfuzz_first_differences"] = [list(map(str, t)) for t in bad[:5]]
        except Exception as e:          # a checker must never take the bench line down
            out["rapidfuzz_error"] = f"{type(e).__name__}: {e}"
    if out["sparse_dot_topn_importable"]:
        try:
            pin = importlib.import_module("pin_sparse_dot_topn")
            bad = rows = 0
            for name, fl, tl, a, b, ntop, lb in pin.cases():
                lib, orc = pin.from_library(a, b, ntop, lb), pin.from_oracle(a, b, ntop, lb)
                for x, y in zip(lib, orc):
                    rows += 1
                    if [c for c, _ in x] != [c for c, _ in y] or any(abs(u[1] - w[1]) > 1e-12 for u, w in zip(x, y)):
                        bad += 1
            out["sparse_dot_topn_rows_differing"] = bad
            out["sparse_dot_topn_checked"] = f"{rows} rows (README lists + 300 x 291 company names at four (ntop, lower_bound) settings)"
        except Exception as e:
            out["sparse_dot_topn_error"] = f"{type(e).__name__}: {e}"
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(live_pins(), indent=1))
