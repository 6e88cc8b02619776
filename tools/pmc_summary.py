#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE).

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM):
the counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64
bytes, so it is doubled; WRITE_SIZE is taken as is.  Pictures per dispatch are
recovered from the grid (row pipelines: BANDS workgroups per picture - 17 for
the bench's 1080p clip, 4 macroblock rows each; k_inter / k_bs / k_residual:
grid.y pictures; convert_batch: grid.z frames)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

BANDS = int(os.environ.get("BANDS", 17))


def load(d, counter):
    out = defaultdict(lambda: [0, 0.0, 0])  # kernel -> [dispatches, value sum, pictures]
    dims = {}
    for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            dims[r["Dispatch_Id"]] = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]),
                                      max(1, int(r["Workgroup_Size_X"])))
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].split("::")[-1]
            gx, gy, gz, wx = dims.get(r["Dispatch_Id"], (int(r["Grid_Size"]), 1, 1, max(1, int(r["Workgroup_Size"]))))
            if name in ("k_intra", "k_deblock", "k_mv"):
                pics = max(1, gx // wx // BANDS)
            elif name == "convert_batch":
                pics = gz
            elif name == "k_upload":
                pics = gy // 6   # kJobUploads arrays per picture
            else:
                pics = gy
            e = out[name]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
            e[2] += pics
    return out


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(set(f) | set(w)):
    fd, fv, fp = f.get(k, [0, 0.0, 0])
    wd, wv, wp = w.get(k, [0, 0.0, 0])
    res[k] = {
        "dispatches": fd or wd, "pictures": fp or wp,
        "fetch_bytes_per_picture": round(2 * fv * 1024 / fp) if fp else None,   # gfx950: FETCH_SIZE x 2
        "write_bytes_per_picture": round(wv * 1024 / wp) if wp else None,
    }
print(json.dumps(res, indent=1))
