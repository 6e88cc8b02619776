#!/bin/bash
# GPU box: instruction mix per kernel (SQ counters), to tell ALU-bound kernels from latency-bound ones.
# Separate --pmc passes, kernel-trace only (gpurun refuses pmc + other trace domains).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
FR=${FRAMES:-10800}
python3 bench.py --frames $FR --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-output --no-reference-api > /dev/null 2>&1
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -o -E "SQ_(WAVES|INSTS_VALU|INSTS_SALU|INSTS_LDS|INSTS_VMEM_RD|INSTS_VMEM_WR|INSTS_VMEM|INSTS_SMEM|WAVE_CYCLES|BUSY_CYCLES|WAIT_INST_ANY|ACTIVE_INST_VALU|INST_CYCLES_VMEM|WAIT_ANY|LDS_BANK_CONFLICT|INSTS_VALU_MFMA_I8)\b" | sort -u | tr '\n' ' ') > gpurun_out/pmc_avail.txt
cat gpurun_out/pmc_avail.txt; echo
i=0
SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT")
for set in "${SETS[@]:0:${NSETS:-3}}"; do
  i=$((i+1)); rm -rf /tmp/pmc_mix$i
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_mix$i -o pmc --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --frames $FR --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-output --no-reference-api > /dev/null 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc_mix$i.err")
  tail -1 gpurun_out/pmc_mix$i.err
done
python3 - <<'PY' > gpurun_out/pmc_instmix.json
import csv, glob, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for d in glob.glob("/tmp/pmc_mix*"):
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            name = r["Kernel_Name"].split("(")[0].split("::")[-1]
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(name, d)].add(r["Dispatch_Id"])
out = {}
for k, c in acc.items():
    w = c.get("SQ_WAVES", 0) or 1
    out[k] = {"dispatches": max(len(v) for (n, d), v in disp.items() if n == k), "waves": c.get("SQ_WAVES")}
    for n, v in c.items():
        if n != "SQ_WAVES":
            out[k][n + "_per_wave"] = round(v / w, 1)
print(json.dumps(out, indent=1))
PY
cat gpurun_out/pmc_instmix.json
