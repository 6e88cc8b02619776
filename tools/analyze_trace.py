#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel, bucketed by grid size
(= batch width), count / mean / total duration; plus the busy fraction of the
traced window per kernel."""
import csv
import glob
import re
import os
import sys
from collections import defaultdict

paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    with open(p) as f:
        rows += list(csv.DictReader(f))
if not rows:
    sys.exit("no kernel_trace.csv under " + sys.argv[1])
# k_upload's grid.y is pictures * kJobUploads (hwang_amd/csrc/amd/frame_job.h): read the constant, do not repeat it
_hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hwang_amd", "csrc", "amd", "frame_job.h")).read()
K_JOB_UPLOADS = int(re.search(r"constexpr int kJobUploads = (\d+);", _hdr).group(1))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
print("kernels %d  window %.3f s" % (len(rows), (t1 - t0) / 1e9))
agg = defaultdict(lambda: [0, 0])
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:40]
    gx, gy = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 1) or 1)
    wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
    jobs = gy if ("k_inter" in name or "k_bs" in name or "k_upload" in name) else max(1, gx // max(wx, 1))
    if "k_upload" in name:
        jobs //= K_JOB_UPLOADS
    b = 1
    while b < jobs:
        b *= 2
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg[(name, b)]
    a[0] += 1
    a[1] += d
tot = defaultdict(int)
for (name, b), (n, d) in sorted(agg.items()):
    print("%-40s jobs<=%-4d n %6d  mean %9.1f us  total %8.3f s" % (name, b, n, d / n / 1e3, d / 1e9))
    tot[name] += d
print()
for name, d in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("%-40s total %8.3f s  (%.1f%% of window)" % (name, d / 1e9, 100.0 * d / (t1 - t0)))

# ---- per-queue busy time and the gaps between consecutive kernels
print()
print("columns:", ",".join(rows[0].keys()))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else None
if qkey:
    byq = defaultdict(list)
    for r in rows:
        byq[r[qkey]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-12:]))
    for q, ev in byq.items():
        ev.sort()
        busy = sum(e - s for s, e, _ in ev)
        span = ev[-1][1] - ev[0][0]
        gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
        big = sorted(gaps)[-5:]
        names = sorted(set(n for _, _, n in ev))
        print("queue %s: kernels %d busy %.3f s span %.3f s (%.0f%%) median gap %.1f us, gaps>1ms: %d (sum %.3f s) kernels=%s" % (
            q, len(ev), busy / 1e9, span / 1e9, 100.0 * busy / max(span, 1), sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0,
            sum(1 for g in gaps if g > 1e6), sum(g for g in gaps if g > 1e6) / 1e9, ",".join(names)))

# ---- a slice of the timeline (all queues), steady state
if len(sys.argv) > 2:
    at = float(sys.argv[2])  # fraction of the window
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 80
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey] if qkey else "?",
                 r["Kernel_Name"].split("(")[0].split("::")[-1][:14],
                 int(r.get("Grid_Size_X", 0)) // max(1, int(r.get("Workgroup_Size_X", 1))), int(r.get("Grid_Size_Y", 1)))
                for r in rows)
    start = t0 + int((t1 - t0) * at)
    sel = [e for e in ev if e[0] >= start][:n]
    print()
    for s, e, q, name, gx, gy in sel:
        print("t=%9.3f ms  q%s  %-14s grid %5d x %4d  dur %8.1f us" % ((s - start) / 1e6, q, name, gx, gy, (e - s) / 1e3))

# ---- the largest idle gaps of the busiest queue, and what ran elsewhere meanwhile
if qkey:
    busiest = max(byq.items(), key=lambda kv: len(kv[1]))[0]
    ev = sorted(byq[busiest])
    gaps = sorted(((ev[i + 1][0] - ev[i][1], ev[i][1], ev[i + 1][0]) for i in range(len(ev) - 1)), reverse=True)[:12]
    print()
    print("largest gaps on queue", busiest)
    for g, gs, ge in sorted(gaps, key=lambda x: x[1]):
        other = defaultdict(int)
        for q, evq in byq.items():
            if q == busiest:
                continue
            for s, e, nme in evq:
                if s < ge and e > gs:
                    other[nme] += 1
        print("  at %8.1f ms  gap %7.2f ms  other queues: %s" % ((gs - t0) / 1e6, g / 1e6, dict(other)))
