#!/usr/bin/env python3
"""median / min of ms_avg per configuration over the repeats of a spmv_sweep.py --repeat run (jsonl on stdin or a file)"""
import json
import sys
from collections import OrderedDict
from statistics import median

rows = OrderedDict()
for line in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    line = line.strip()
    if not line.startswith("{"):
        continue
    r = json.loads(line)
    rows.setdefault(r["config"], []).append(r)
for name, rs in rows.items():
    ms = [r["ms_avg"] for r in rs]
    print("%-14s n=%d median %.4f min %.4f max %.4f  frac(median) %.4f  %s" % (name, len(ms), median(ms), min(ms), max(ms),
          rs[0]["frac_of_8TBs"] * rs[0]["ms_avg"] / median(ms), json.dumps(rs[0]["opts"])))
