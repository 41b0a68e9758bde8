#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd sqlite outputs (kernel-trace stats and PMC passes) as text.

usage: rocprof_summary.py <dir-with-*_results.db files ...>
"""
import glob
import os
import sqlite3
import sys


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    out = [f"## kernel-trace stats: {db}", f"{'calls':>6} {'total_us':>14} {'avg_us':>12} {'pct':>7}  name"]
    for name, calls, tot, avg, pct in rows:
        out.append(f"{calls:6d} {tot:14.3f} {avg:12.3f} {pct:7.3f}  {name[:110]}")
    return "\n".join(out)


# (round 5: "bb_" and "gz_" were missing from this list until now -- the Bloom partition kernels' and the gzip decoder's counter rows
# were collected in round 4 and dropped HERE, which profiles/README.md then reported as "returned no rows on this pool")
def pmc_stats(db, kernel_filter=("rows_kernel", "count27", "seq_kernel", "inflate", "fq_", "hmm_", "bloom_", "cov_kernel", "node_gather", "table_", "bb_",
                                 "gz_", "even_debit", "countkc", "ct_", "ctd_", "xt_", "pt_")):      # (round 6: "ctd_" added -- the deferred pass's two kernels were dropped here at first)
    cur = sqlite3.connect(db).cursor()
    try:
        rows = cur.execute(
            "select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection "
            "group by kernel_name, counter_name").fetchall()
    except sqlite3.OperationalError:
        return ""
    out = [f"## PMC: {db}", f"{'dispatches':>10} {'sum':>22} {'per_dispatch':>22}  counter  kernel"]
    for kn, cn, nd, s in rows:
        if kernel_filter and not any(f in kn for f in kernel_filter):
            continue
        out.append(f"{nd:10d} {s:22.1f} {s / max(nd, 1):22.1f}  {cn}  {kn[:60]}")
    return "\n".join(out) if len(out) > 2 else ""


def main():
    for d in sys.argv[1:]:
        for db in sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)):
            cur = sqlite3.connect(db).cursor()
            n = cur.execute("select count(*) from counters_collection").fetchone()[0]
            txt = pmc_stats(db) if n else kernel_stats(db)
            if txt:
                print(txt)
                print()


if __name__ == "__main__":
    main()
