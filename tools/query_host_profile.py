"""Where the HOST time of a single-query TFIDF.match goes (cProfile over 3 000 queries against the fitted 100k list)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyfuzz_amd import datasets
from polyfuzz_amd.models import TFIDF
names = datasets.load_company_names()
m = TFIDF(min_similarity=0, top_n=1)
m.match(names[:1000], names)
q = [names[50000]]
for _ in range(50): m.match(q, names, re_train=False)
t0 = time.perf_counter()
for _ in range(3000): m.match(q, names, re_train=False)
print("wall per query %.4f ms" % ((time.perf_counter() - t0) / 3000 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): m.match(q, names, re_train=False)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
