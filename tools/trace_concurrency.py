#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and prints per-kernel stats plus how many
kernels overlap in time (concurrency across streams)."""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
if not rows:
    print("no kernel trace found")
    sys.exit(0)
ev = []
stat = defaultdict(lambda: [0, 0.0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-40:]
    stat[name][0] += 1
    stat[name][1] += (e - s) / 1e3
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
busy = defaultdict(float)
cur, last = 0, ev[0][0]
for t, d in ev:
    busy[cur] += t - last
    cur += d
    last = t
span = (t1 - t0) / 1e6
print("span %.1f ms, %d kernels" % (span, len(rows)))
for k in sorted(busy):
    print("  %2d kernels in flight: %6.1f ms (%.0f%%)" % (k, busy[k] / 1e6, 100 * busy[k] / (t1 - t0)))
for n, (c, us) in sorted(stat.items(), key=lambda x: -x[1][1]):
    print("%-42s n=%6d total %9.1f ms avg %8.1f us" % (n, c, us / 1e3, us / c))
