#!/usr/bin/env python3
"""Per-dispatch view of a rocprofv3 --kernel-trace result (rocpd SQLite): the kernels whose name contains one
of the given substrings, grouped by (name, grid), with mean / min duration over the dispatches, and the
dispatch sequence of the last repetition with the gaps between kernels.
usage: rocprof_seq.py <results.db> [start=<substring of the kernel that opens a repetition; default permute>] substring [substring ...]"""
import sqlite3
import sys
from collections import OrderedDict


def short(name, n=84):
    name = name.replace("void ", "").replace("sprs_hip::", "").replace("(anonymous namespace)::", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    args = sys.argv[2:]
    opener = "permute"
    for a in list(args):
        if a.startswith("start="):
            opener = a[6:]
            args.remove(a)
    subs = args or ["sprs_hip"]
    rows = [r for r in db.execute("select name, grid_x, workgroup_x, start, end from kernels order by start")
            if any(s in r[0] for s in subs)]
    groups = OrderedDict()
    for name, gx, wx, st, en in rows:
        groups.setdefault((name, gx, wx), []).append(en - st)
    print("# per (kernel, grid): dispatches, mean us, min us")
    print("%-86s %10s %6s %6s %10s %10s" % ("kernel", "grid", "wg", "n", "mean_us", "min_us"))
    for (name, gx, wx), d in groups.items():
        print("%-86s %10d %6d %6d %10.2f %10.2f" % (short(name), gx // max(wx, 1), wx, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3))
    # last repetition: walk back from the end until the first group's kernel name repeats
    if rows:
        first_name = None
        seq = []
        for r in reversed(rows):
            seq.append(r)
            if first_name is None:
                first_name = r[0]
            if opener in r[0]:
                break
        seq.reverse()
        print("# last repetition (start-to-start offsets in us)")
        t0 = seq[0][3]
        prev_end = None
        for name, gx, wx, st, en in seq:
            gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
            print("  +%9.2f  dur %9.2f  gap %6.2f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, gap, short(name, 70)))
            prev_end = en
        print("  total span %.2f us" % ((seq[-1][4] - t0) / 1e3))


if __name__ == "__main__":
    main()
