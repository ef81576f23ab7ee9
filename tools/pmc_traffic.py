"""rocprofv3 PMC passes of tools/pmc_traffic.sh -> the JSON bench.py reads as profiles/k3_hbm_traffic.json.
usage: python tools/pmc_traffic.py <outdir of pmc_traffic.sh> > profiles/k3_hbm_traffic.json"""
import json
import os
import sqlite3
import sys

out_dir = sys.argv[1]
KEYS = {"tfidf": "headline", "c2": "c2_tfidf_10k", "tfidf_1m": "tfidf_1m_shard"}
WHAT = {"tfidf": "company_names[:100000] self-match top-5", "c2": "config 2: 10 000 x 10 000 company names, top-5",
        "tfidf_1m": "one GPU's shard of config 4: 125 000 x 1 000 000 synthetic names, top-10"}


def avg(db_path, counter):
    """(kernel name, K3 launches, average per K3 launch) of the K3 form that ran: the row-major or the lock-step kernel (one
    dispatch per launch), or -- a list against itself -- the symmetric form, whose launch is four dispatches (k3_sym_kernel
    in its three passes + k3_sym_order / repost / merge / merge_slices: summed, per dispatch of the merge)"""
    db = sqlite3.connect(db_path)
    sym = db.execute("select count(*) from counters_collection where counter_name = ? and (kernel_name like '%k3_sym_merge<%' or kernel_name like '%k3_sym_merge(%')",
                     (counter,)).fetchone()[0]
    if sym:
        total = db.execute("select sum(value) from counters_collection where counter_name = ? and kernel_name like '%k3_sym_%'",
                           (counter,)).fetchone()[0]
        return ("pfz::k3_sym_kernel<2048> x3 passes + k3_sym_order / repost / merge / merge_slices", sym, total / sym)
    rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? and "
                      "(kernel_name like '%k3_cossim_topn_kernel%' or kernel_name like '%k3_lockstep_kernel%') group by kernel_name order by sum(value) desc", (counter,)).fetchall()
    return rows[0] if rows else (None, 0, None)


records = {}
for cfg, key in KEYS.items():
    f = os.path.join(out_dir, f"{cfg}_FETCH_SIZE", "bench_results.db")
    w = os.path.join(out_dir, f"{cfg}_WRITE_SIZE", "bench_results.db")
    if not (os.path.exists(f) and os.path.exists(w)):
        continue
    kf, nf, fetch = avg(f, "FETCH_SIZE")
    kw, nw, write = avg(w, "WRITE_SIZE")
    if fetch is None or write is None:
        continue
    records[key] = {
        "kernel": kf.split("(")[0].replace("void ", ""), "workload": WHAT[cfg],
        "source": f"rocprofv3 --pmc FETCH_SIZE ({nf} launches) / --pmc WRITE_SIZE ({nw} launches), separate passes, "
                  f"tools/pmc_traffic.sh: python bench.py --config {cfg}",
        "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
        "correction": "x2 on FETCH_SIZE (MI355X_MICROARCH.md 'HBM'; tools/ubench/fetch_calib.hip: K3's 8-byte-per-lane loads "
                      "report 0.500 of the bytes read, like 16-byte-per-lane loads)",
        "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
        "note": "bytes per K3 launch that left L2 for the fabric (FETCH_SIZE x2 + WRITE_SIZE); Infinity-Cache hits are counted, so "
                "this is an upper bound of HBM bytes",
    }
json.dump({"what": "bytes per K3 launch that left L2 (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; "
                   "FETCH_SIZE x2 per /opt/skills/guides/MI355X_MICROARCH.md 'HBM' and the calibration of "
                   "tools/ubench/fetch_calib.hip); written by tools/pmc_traffic.py, read by bench.py (roofline.traffic)",
           "records": records}, sys.stdout, indent=1)
