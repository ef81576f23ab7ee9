import sys, time; sys.path.insert(0,'.')
import polyfuzz_amd
from polyfuzz_amd import pipeline, datasets
ctx = polyfuzz_amd.Context(0)
names = datasets.load_company_names()
job = pipeline.TfidfMatchJob(ctx, names, None, top_n=5, min_similarity=0.0, self_match=True)
for _ in range(3): job.step()
ctx.sync()
for prof in (False, True, False, True):
    ctx.prof_enable(prof); ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(20): job.step()
    ctx.sync()
    print("prof", prof, "ms/step", (time.perf_counter() - t0) / 20 * 1e3)
