"""the device timeline of the second-to-last TFIDF.match of tools/r5_match_trace.sh's rocprofv3 run (gpurun_out/r5_match_trace/t/*.db)"""
import sqlite3, glob, sys
db = sqlite3.connect(glob.glob('gpurun_out/r5_match_trace/t/*.db')[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
cp = db.execute("select name, start, end from memory_copies order by start").fetchall()
ev = sorted([(s, e, n[:60]) for n, s, e in rows] + [(s, e, 'COPY ' + str(n)[:40]) for n, s, e in cp])
starts = [i for i in range(len(ev)) if 'HOST_TO_DEVICE' in ev[i][2]]
ms = [starts[0]]
for i in starts[1:]:
    if ev[i][0] - ev[ms[-1]][0] > 2e6:
        ms.append(i)
a, b = ms[-3], ms[-2]
t0 = ev[a][0]
skip = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for s, e, n in ev[a:b]:
    if (s - t0) / 1e3 >= skip:
        print('%8.1f %7.1f us  end %8.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, n))
