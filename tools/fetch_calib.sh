#!/bin/bash
# GPU box: FETCH_SIZE of tools/experiments/fetch_calib.hip's kernels (known byte counts, the access shapes of k_deblock
# and k_inter) -> gpurun_out/fetch_calib.txt: per kernel the counter (KiB), the buffer's bytes and their ratio.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/experiments/fetch_calib.hip -o /tmp/fetch_calib || exit 1
rm -rf /tmp/fc
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/fc -o calib --output-format csv -- /tmp/fetch_calib > /tmp/fc_stdout.txt 2>&1)
python3 - <<'PY' > gpurun_out/fetch_calib.txt
import csv, glob
rows = []
for p in glob.glob("/tmp/fc/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(p)) if r.get("Counter_Name") == "FETCH_SIZE"]
per = {}
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    per.setdefault(name, []).append(float(r["Counter_Value"]))
n = 1 << 30
print("kernel       dispatches  FETCH_SIZE KiB (2nd dispatch)  x1024 / buffer bytes   -> factor that turns the counter into bytes touched")
for k, v in per.items():
    c = v[-1] * 1024
    print("%-12s %d           %14.0f                  %.3f                  %.2f" % (k, len(v), v[-1], c / n, n / c if c else 0))
PY
cat gpurun_out/fetch_calib.txt
