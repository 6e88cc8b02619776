#!/bin/bash
# GPU box: the default bench line, then the same command under rocprofv3
# (--kernel-trace --stats); summaries land in gpurun_out/ for profiles/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
tail -3 gpurun_out/bench.err
rm -rf /tmp/prof_bench
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-hbm-output --no-reference-api "$@" > "$GRAFT_REPO_ROOT/gpurun_out/bench_profiled.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/bench_profiled.err")
cat gpurun_out/bench_profiled.json
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/bench_kernel_stats.csv \;
find /tmp/prof_bench -name "*domain_stats.csv" -exec cp {} gpurun_out/bench_domain_stats.csv \;
cat gpurun_out/bench_kernel_stats.csv
python3 tools/analyze_trace.py /tmp/prof_bench > gpurun_out/bench_trace_summary.txt 2>&1
tail -12 gpurun_out/bench_trace_summary.txt
