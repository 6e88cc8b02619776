#!/bin/bash
# GPU box: HBM traffic of the HEVC kernels from the L2's memory-side counters - two separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE) of a shortened tools/hevc_bench.py run, as tools/pmc_profile.sh does for the H.264 kernels.
# Per-kernel bytes per picture -> gpurun_out/r06_hevc_pmc_summary.json
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 tools/hevc_bench.py --frames 7200 --steps 1 > /dev/null 2>&1   # the clip
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/hpmc_$c
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d /tmp/hpmc_$c -o pmc --output-format csv -- python3 "$GRAFT_REPO_ROOT/tools/hevc_bench.py" --frames 7200 --steps 1 > "$GRAFT_REPO_ROOT/gpurun_out/hpmc_$c.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/hpmc_$c.err")
  tail -2 gpurun_out/hpmc_$c.err
done
python3 - /tmp/hpmc_FETCH_SIZE /tmp/hpmc_WRITE_SIZE > gpurun_out/r06_hevc_pmc_summary.json <<'PY'
import csv, glob, json, re, sys
from collections import defaultdict
def load(d, counter):
    out = defaultdict(lambda: [0, 0.0, 0])
    dims = {}
    for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            dims[r["Dispatch_Id"]] = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), max(1, int(r["Workgroup_Size_X"])), max(1, int(r["Workgroup_Size_Y"])), max(1, int(r["Workgroup_Size_Z"])))
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r.get("Counter_Name") != counter:
                continue
            m = re.search(r"k_hevc_\w+", r["Kernel_Name"])
            if not m:
                continue
            name = m.group(0)
            gx, gy, gz, wx, wy, wz = dims.get(r["Dispatch_Id"], (0, 1, 1, 1, 1, 1))
            by, bz = gy // wy, gz // wz
            pics = by // 3 if name == "k_hevc_sao" else bz if name in ("k_hevc_intra_rows", "k_hevc_intra_flow") else by
            e = out[name]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
            e[2] += max(1, pics)
    return out
f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(set(f) | set(w)):
    fd, fv, fp = f.get(k, [0, 0.0, 0])
    wd, wv, wp = w.get(k, [0, 0.0, 0])
    # MI355X_MICROARCH.md (HBM): the counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes: doubled
    res[k] = {"dispatches": fd or wd, "pictures": fp or wp, "fetch_bytes_per_picture": round(2 * fv * 1024 / fp) if fp else None,
              "write_bytes_per_picture": round(wv * 1024 / wp) if wp else None}
print(json.dumps(res, indent=1))
PY
cat gpurun_out/r06_hevc_pmc_summary.json
