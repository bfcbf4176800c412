#!/usr/bin/env python3
"""The last `window_ms` of a rocprofv3 kernel-trace CSV as a timeline summary: per kernel (and grid size) the launches,
average duration and share; the time the device ran at least one kernel against the window (what is left is launch
gaps, host waits and exchanges).  Usage: trace_tail.py <kernel_trace.csv> <window_ms> [list]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:],
             int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) for r in rows)
t1 = max(e[1] for e in ev)
ev = [e for e in ev if e[0] >= t1 - win]
t0 = ev[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window %.3f ms, %d launches, device busy %.3f ms (%.1f %%)" % ((t1 - t0) / 1e6, len(ev), busy / 1e6, 100. * busy / (t1 - t0)))
d = collections.defaultdict(list)
for s, e, k, b in ev:
    d[(k, b)].append((e - s) / 1e3)
for (k, b), v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print("%-60s %8d %6d x %8.1f us = %8.3f ms" % (k, b, len(v), sum(v) / len(v), sum(v) / 1e3))
if len(sys.argv) > 3:
    prev = t0
    for s, e, k, b in ev[-int(sys.argv[3]):]:
        print("%9.1f gap %7.1f dur %7.1f  %s %d" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, k[-40:], b))
        prev = max(prev, e)
