#!/bin/bash
# GPU box: everything profiles/rNN_* is made from, in one call:
#   the -m gpu suite, the default bench line + the same command under rocprofv3 (tools/bench_profile.sh), the two PMC
#   passes (tools/pmc_profile.sh), and the C2 / C4 bench lines.  Results land in gpurun_out/.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python3 -m pytest tests -x -q -m gpu > gpurun_out/gputest.txt 2>&1; tail -3 gpurun_out/gputest.txt
bash tools/bench_profile.sh > gpurun_out/bench_profile.log 2>&1; head -c 600 gpurun_out/bench.json; echo
bash tools/pmc_profile.sh > gpurun_out/pmc_profile.log 2>&1; tail -5 gpurun_out/pmc_profile.log
for c in C2 C4; do
  python3 bench.py --config $c --no-reference-api > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
  head -c 400 gpurun_out/bench_$c.json; echo
done
