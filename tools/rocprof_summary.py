"""Summarise rocprofv3 rocpd databases (gpurun_out/...) into small text files under profiles/.
usage: python tools/rocprof_summary.py <db> [<db> ...] > profiles/<name>.txt"""
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    print(f"== {path}")
    rows = db.execute("select * from top_kernels").fetchall()
    if rows:
        print("kernel-trace --stats: name | calls | total_us | avg_us | pct")
        for r in rows[:24]:
            print(f"  {r[0][:60]:60s} {r[1]:6d} {r[2]:14.1f} {r[3]:12.2f} {r[4]:6.2f}")
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
         "group by kernel_name, counter_name order by kernel_name, counter_name")
    try:
        cur = db.execute(q).fetchall()
    except sqlite3.Error:
        cur = []
    if cur:
        print("pmc (average per dispatch): kernel | counter | dispatches | avg")
        for r in cur:
            if any(k in r[0] for k in ("k3_", "k4_", "k_extract", "k_rows", "k5_", "k7_")):
                print(f"  {r[0][:44]:44s} {r[1]:24s} {r[2]:5d} {r[3]:18.1f}")
