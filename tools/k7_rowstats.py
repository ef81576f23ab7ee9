"""Where K7's time goes, row by row (PFZ_K7_ROW_STATS): pairs scored and clock ticks of every from-title of config 3's lists,
against its length / tokens; and the kernel with scoring switched off (PFZ_K7_EXP=1: the two bounding sweeps alone)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polyfuzz_amd
from polyfuzz_amd import _lib, datasets
ctx = polyfuzz_amd.Context.default()
fl, tl = datasets.c3_lists()
mode = sys.argv[1] if len(sys.argv) > 1 else "WRatio"
f_dev, t_dev = _lib.DeviceStrings.upload(ctx, fl), _lib.DeviceStrings.upload(ctx, tl)
out = _lib.DeviceTopN.alloc(ctx, len(fl), 2)
def run(label, reps=3):
    _lib.fuzz_extract_one_dev(ctx, f_dev, t_dev, mode, out); ctx.sync()
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(reps): _lib.fuzz_extract_one_dev(ctx, f_dev, t_dev, mode, out)
    ctx.sync(); k_ms, _ = ctx.prof_get("k7_fuzz"); ctx.prof_enable(False)
    print(f"{label}: k7_fuzz {k_ms / reps:.2f} ms")
run("as shipped")
os.environ["PFZ_K7_EXP"] = "1"; run("bounds only (no scoring)"); del os.environ["PFZ_K7_EXP"]
path = "/tmp/k7_stats.bin"
os.environ["PFZ_K7_ROW_STATS"] = path
_lib.fuzz_extract_one_dev(ctx, f_dev, t_dev, mode, out); ctx.sync()
del os.environ["PFZ_K7_ROW_STATS"]
raw = np.fromfile(path, np.uint64)
n = (len(raw) - 24) // 4
ext = raw[2 * n:2 * n + 24].astype(np.float64)
phase = ext[:7]
st = raw[:2 * n].reshape(-1, 2).astype(np.float64)
span = raw[2 * n + 24:].reshape(-1, 2)
t0 = ((1 << 62) - span[:, 0].astype(np.int64)).astype(np.float64); t1 = span[:, 1].astype(np.float64)
k0 = t0.min(); t0 = (t0 - k0) / 100.0; t1 = (t1 - k0) / 100.0           # microseconds since the first unit began (100 MHz clock)
print(f"timeline: last row ends at {t1.max():.0f} us; rows still running at 50/60/70/80/90/95 % of that:",
      [int(((t0 <= f * t1.max()) & (t1 > f * t1.max())).sum()) for f in (0.5, 0.6, 0.7, 0.8, 0.9, 0.95)])
late = np.argsort(-t1)[:10]
names = ["set-up of a from-string", "sweep 1 (bounds, seeds)", "sweep 2 (cached bounds)", "scoring batches", "end of unit / next unit", "waiting for a continuation", "window sweeps"]
print("wave time by phase (shader clock, all waves):", " | ".join(f"{n} {phase[i] / phase.sum():.3f}" for i, n in enumerate(names)), f"| total {phase.sum():.3e} ticks")
b, act, go, sw, win, paid = ext[8:14]
if ext[16:22].sum() > 0:          # (a -DPFZ_K7_PROFILE build)
    sub = ext[16:22]
    print("scoring, slowest lane of each batch by sub-phase (share of the scoring ticks", f"{sub.sum() / phase[3]:.2f}):",
          " | ".join(f"{n} {x / sub.sum():.3f}" for n, x in zip(["tokens/set-up", "LCS passes", "token-set pass", "window sweeps", "metadata + refined bound", "rest"], sub)))
print(f"scoring batches {b:.0f}: lanes with a pair {act / (64 * b):.3f}, scored {go / (64 * b):.3f}, swept windows {sw / (64 * b):.3f}; windows per sweeping lane {win / max(sw, 1):.2f}, "
      f"windows the waves paid for / windows swept {paid / max(win, 1):.2f}")
scored, ticks = st[:, 0], st[:, 1]
L = np.array([len(s) for s in fl]); T = np.array([len(set(s.split())) for s in fl])
idx, score = _lib.best_from_topn(*out.download())
print("ticks per row: pct", np.percentile(ticks, [1, 25, 50, 75, 90, 99, 100]).round(0), "sum", ticks.sum(), "(100 MHz clock: 1 tick = 10 ns)")
print("scored per row: pct", np.percentile(scored, [1, 25, 50, 75, 90, 99, 100]).round(0))
for lo, hi in ((0, 4), (5, 8), (9, 12), (13, 16), (17, 24), (25, 32), (33, 64), (65, 999)):
    m = (L >= lo) & (L <= hi)
    if m.any():
        print(f"len {lo:3d}-{hi:3d}: rows {m.sum():6d}  scored/row {scored[m].mean():9.0f}  us/row {ticks[m].mean() / 100:9.1f}  share of time {ticks[m].sum() / ticks.sum():.3f}  mean best {score[m].mean():.1f}")
order = np.argsort(-ticks)[:12]
for i in order: print(f"  {ticks[i] / 100:9.1f} us  scored {scored[i]:7.0f}  best {score[i]:.1f}  {fl[i]!r}")
print("the rows that end last:")
for i in late: print(f"  began {t0[i]:8.0f} us  ended {t1[i]:8.0f} us  scored {st[i, 0]:7.0f}  len {len(fl[i]):3d}  {fl[i]!r}")
print("corr(ticks, scored)", np.corrcoef(ticks, scored)[0, 1])
