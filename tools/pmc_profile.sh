#!/bin/bash
# GPU box: HBM traffic of the kernels from the L2's memory-side counters, two
# separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one
# pass on gfx950), on a shortened bench run (the profiler serialises
# dispatches).  Per-kernel sums -> gpurun_out/pmc_summary.json.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
FR=${FRAMES:-10800}
python3 bench.py --frames $FR --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-output --no-reference-api > /dev/null 2>&1   # corpus + warm caches
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o pmc --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --frames $FR --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-output --no-reference-api > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c.err")
  tail -2 gpurun_out/pmc_$c.err
done
python3 tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > gpurun_out/pmc_summary.json
cat gpurun_out/pmc_summary.json
