#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) result for profiles/:
top kernels by total time, and — when the run collected PMC counters — the
per-kernel mean of every counter for the sprs_hip kernels.
usage: rocprof_summary.py <results.db> [kernel-substring]"""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    filt = sys.argv[2] if len(sys.argv) > 2 else "sprs_hip"
    cur = db.cursor()
    print("# kernel-trace --stats (rocprofv3), source: %s" % sys.argv[1].split("/")[-1])
    print("%-112s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    shown = 0
    for name, calls, tot, avg, pct in rows:
        if filt in name or shown < 6:
            print("%-112s %8d %14.1f %12.2f %7.2f" % (short(name), calls, tot / 1e3 if tot > 1e7 else tot, avg / 1e3 if tot > 1e7 else avg, pct))
            shown += 1
    try:
        q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
             "from counters_collection where kernel_name like ? group by kernel_name, counter_name")
        pm = list(cur.execute(q, ("%" + filt + "%",)))
    except sqlite3.Error:
        pm = []
    if pm:
        print("\n# PMC counters per dispatch (mean over dispatches)")
        print("%-70s %-28s %6s %16s %16s %16s" % ("kernel", "counter", "n", "mean", "min", "max"))
        for k, c, n, a, lo, hi in pm:
            print("%-70s %-28s %6d %16.6g %16.6g %16.6g" % (short(k, 70), c, n, a, lo, hi))


if __name__ == "__main__":
    main()
