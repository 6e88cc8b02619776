#!/bin/bash
# GPU box: everything profiles/rNN_* is made from, in one call:
#   the -m gpu suite, the default bench line + the same command under rocprofv3 (tools/bench_profile.sh), the two PMC
#   passes (tools/pmc_profile.sh), and the C2 / C4 bench lines.  Results land in gpurun_out/.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1800 python3 -u -m pytest tests -x -q -m gpu --timeout 600 > gpurun_out/gputest.txt 2>&1; tail -3 gpurun_out/gputest.txt
bash tools/bench_profile.sh > gpurun_out/bench_profile.log 2>&1; head -c 600 gpurun_out/bench.json; echo
bash tools/pmc_profile.sh > gpurun_out/pmc_profile.log 2>&1; tail -5 gpurun_out/pmc_profile.log
timeout 600 python3 tools/replay_probe.py frames=10800 widths=8,48,111 launches=40 out=gpurun_out/replay.json > /dev/null 2> gpurun_out/replay.err; tail -2 gpurun_out/replay.err
timeout 600 python3 tools/replay_probe.py frames=10800 lanes=128 widths=48,128,185 launches=30 out=gpurun_out/replay_128lanes.json > /dev/null 2> gpurun_out/replay128.err; tail -2 gpurun_out/replay128.err
NSETS=2 bash tools/pmc_instmix.sh > gpurun_out/pmc_instmix.log 2>&1; tail -3 gpurun_out/pmc_instmix.log
for c in C2 C4; do
  python3 bench.py --config $c --no-reference-api > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
  head -c 400 gpurun_out/bench_$c.json; echo
done
