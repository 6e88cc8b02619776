# the round's closing evidence on one box: GPU tests, smoke, the driver's bench command, kernel stats of the same command
set -x
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06_final_gpu_tests.txt 2>&1; tail -3 gpurun_out/r06_final_gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_final_smoke.txt 2>&1; tail -2 gpurun_out/r06_final_smoke.txt
timeout 1500 python bench.py > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err; tail -c 300 gpurun_out/r06_final_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o final --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-reference-api --no-hevc > $GRAFT_REPO_ROOT/gpurun_out/r06_final_bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_final_prof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_final_bench_kernel_stats.csv && head -14 gpurun_out/r06_final_bench_kernel_stats.csv
rm -rf gpurun_out/prof_final
