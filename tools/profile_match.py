"""Host-side profile of TFIDF.match on Python lists (where the wall time of the drop-in call goes)."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyfuzz_amd import synth
from polyfuzz_amd.models import TFIDF

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
tl = synth.company_names(n, 5678)
fl = synth.company_names(n, 1234)
m = TFIDF(n_gram_range=(3, 3), top_n=5, min_similarity=0.0)
m.match(fl, tl)
t0 = time.perf_counter(); m.match(fl, tl); print("match wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); df = m.match(fl, tl); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
print(df.head(3))
from polyfuzz_amd import _lib, pipeline
ctx = _lib.Context.default()
ctx.sync(); ctx.prof_enable(True); ctx.prof_reset()
t0 = time.perf_counter(); m.match(fl, tl); ctx.sync(); dt = time.perf_counter() - t0
print("kernels (ms):", {k: round(ctx.prof_get(k)[0], 3) for k in pipeline.PROFILED_KERNELS}, "wall %.1f ms" % (dt * 1e3))
ctx.prof_enable(False)
