"""Experiment: K3 with the to-rows ordered by the norm of their heavy part (tight per-block pruning bounds).

The to-list is permuted on the host here; a production version would permute inside pfz_index_build.
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polyfuzz_amd
from polyfuzz_amd import synth, pipeline

H = int(os.environ.get("PFZ_K3_HEAVY", "32"))
ctx = polyfuzz_amd.Context.default()
tl = synth.company_names(100_000, 5678)
fl = synth.company_names(100_000, 1234)


def k3_ms(job, steps=8):
    for _ in range(2):
        job.step()
    ctx.sync(); ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(steps):
        job.step()
    ctx.sync(); ms, n = ctx.prof_get("k3_cossim_topn"); ctx.prof_enable(False)
    return ms / n


job = pipeline.TfidfMatchJob(ctx, fl, tl, top_n=5, min_similarity=0.0)
print(f"as generated          k3 {k3_ms(job):.3f} ms", flush=True)
a3, b3, n_col = job.host_matrices()
indptr, indices, data = b3
df = np.bincount(indices, minlength=n_col)
heavy = np.zeros(n_col, bool)
heavy[np.argsort(-df, kind="stable")[:H]] = True
w = np.where(heavy[indices], data.astype(np.float64) ** 2, 0.0)
hn = np.sqrt(np.add.reduceat(np.concatenate([w, [0.0]]), indptr[:-1].astype(np.int64)) * (np.diff(indptr) > 0))
for name, order in (("identity", np.arange(len(tl))),
                    ("heavy-norm descending", np.argsort(-hn, kind="stable")),
                    ("heavy-norm ascending", np.argsort(hn, kind="stable"))):
    tl2 = [tl[i] for i in order]
    job2 = pipeline.TfidfMatchJob(ctx, fl, tl2, top_n=5, min_similarity=0.0)
    print(f"{name:22s} k3 {k3_ms(job2):.3f} ms", flush=True)
    if os.environ.get("PFZ_K3_STATS"):
        pass
