#!/usr/bin/env python3
"""Summarises a rocprofv3 kernel-trace CSV per (kernel, grid size): calls, average/min/max duration.
rocprofv3 --stats averages a kernel symbol over all its launches; the multigrid hierarchy launches
the same symbol on every level, so the finest-level figures the roofline uses are only visible
per grid size.  Usage: summarize_trace.py <kernel_trace.csv> [min_total_ms]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
d = collections.defaultdict(list)
for r in rows:
    blocks = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    d[(r["Kernel_Name"].split("(")[0], blocks)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-70s %9s %7s %10s %10s %10s %10s" % ("kernel", "blocks", "calls", "avg_us", "min_us", "max_us", "total_ms"))
for (k, b), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / 1e3 >= thr:
        print("%-70s %9d %7d %10.1f %10.1f %10.1f %10.2f" % (k[-70:], b, len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1e3))
