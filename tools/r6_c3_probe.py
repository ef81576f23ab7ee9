"""round 6: where the host time of EditDistance.match / RapidFuzz.match goes on config 3's lists (20 000 x 20 000 IMDB titles)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cProfile, pstats
from polyfuzz_amd import datasets
from polyfuzz_amd.models import EditDistance, RapidFuzz
fl, tl = datasets.c3_lists(20000)
for cls in (EditDistance, RapidFuzz):
    m = cls()
    for _ in range(4): m.match(fl, tl)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); m.match(fl, tl); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort(); print(cls.__name__, "match median %.3f ms" % ts[7], getattr(m, "last_timings", None))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): m.match(fl, tl)
    pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(10)
