"""Per-stream occupancy of the HEVC engine from a rocprofv3 kernel trace (tools/experiments/r06_hevc_kernel_trace.sh).

For every HIP stream: how many kernels, the fraction of the busy span (first to last kernel of the steady part) in which one of
its kernels ran, and the gaps between consecutive kernels.  And per "batch" (a k_hevc_clear .. k_hevc_sao run on one stream):
pictures (grid y), duration, and the gap to the next batch of the stream."""
import csv
import gzip
import sys
from collections import defaultdict

rows = list(csv.DictReader(gzip.open(sys.argv[1], "rt")))
t_all0 = min(int(r["Start_Timestamp"]) for r in rows)
per = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    short = name.split("(")[0].split("::")[-1] if not name.startswith("(") else name.split("::", 1)[1].split("(")[0]
    per[r["Stream_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r.get("Grid_Size_Y") or 0) // max(1, int(r.get("Workgroup_Size_Y") or 1)),
                                int(r.get("Grid_Size_Z") or 0)))
# the steady part: the last 60 % of the trace (the first call builds sets and surfaces)
t_end = max(int(r["End_Timestamp"]) for r in rows)
t_from = t_all0 + int(0.4 * (t_end - t_all0))
print("trace %.2f s; looking at the last %.2f s" % ((t_end - t_all0) / 1e9, (t_end - t_from) / 1e9))
for stream, ks in sorted(per.items(), key=lambda kv: -len(kv[1])):
    ks = sorted(k for k in ks if k[0] >= t_from)
    if len(ks) < 10:
        continue
    busy = sum(e - s for s, e, *_ in ks)
    span = ks[-1][1] - ks[0][0]
    gaps = sorted(ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1))
    names = defaultdict(int)
    for k in ks:
        names[k[2]] += 1
    print("stream %s: %d kernels, busy %.3f of %.2f s; gap between kernels: median %.1f us, 90 %% %.1f us, max %.1f us; %s" % (
        stream, len(ks), busy / span, span / 1e9, gaps[len(gaps) // 2] / 1e3, gaps[int(0.9 * len(gaps))] / 1e3, gaps[-1] / 1e3,
        ", ".join("%s %d" % kv for kv in sorted(names.items()))))
    # batches: from a k_hevc_clear / k_hevc_residual that follows a k_hevc_sao (or the start) to the next k_hevc_sao
    batches = []
    cur = None
    for s, e, name, gy, gz in ks:
        if not name.startswith("k_hevc"):
            continue
        if cur is None:
            cur = [s, e, 0]
        cur[1] = e
        if name == "k_hevc_residual":
            cur[2] = gy
        if name == "k_hevc_sao":
            batches.append(tuple(cur))
            cur = None
    if len(batches) > 4:
        dur = sorted(b[1] - b[0] for b in batches)
        pics = [b[2] for b in batches]
        idle = sorted(batches[i + 1][0] - batches[i][1] for i in range(len(batches) - 1))
        print("   %d batches, %.2f pictures each (max %d); duration median %.2f ms, 90 %% %.2f ms; idle before the next: median %.2f ms, 90 %% %.2f ms, total %.3f of the span"
              % (len(batches), sum(pics) / len(pics), max(pics), dur[len(dur) // 2] / 1e6, dur[int(0.9 * len(dur))] / 1e6, idle[len(idle) // 2] / 1e6,
                 idle[int(0.9 * len(idle))] / 1e6, sum(idle) / span))

# how a kernel's duration depends on the pictures in its launch
by = defaultdict(list)
for r in rows:
    if int(r["Start_Timestamp"]) < t_from or "k_hevc" not in r["Kernel_Name"] or "deblock" in r["Kernel_Name"]:
        continue   # (k_hevc_deblock's name has a comma: the trace was cut by columns)
    s = r["Kernel_Name"].split("::", 1)[1].split("(")[0]
    gy = int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"])
    gz = int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_Z"])
    k = gz if s in ("k_hevc_intra_flow", "k_hevc_intra_rows") else gy // 3 if s == "k_hevc_sao" else gy
    by[(s, k)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for s in ["k_hevc_intra_rows", "k_hevc_intra_flow", "k_hevc_residual", "k_hevc_inter", "k_hevc_sao"]:
    print(s + ": pictures in the launch -> launches, median us")
    print("   " + "; ".join("%d -> %d, %.0f" % (k, len(v), sorted(v)[len(v) // 2]) for (n, k), v in sorted(by.items()) if n == s and len(v) >= 5))
