set -x
timeout 900 python -m pytest tests/test_gpu_hevc_kernels.py tests/test_host_hevc.py -x -q -m gpu > gpurun_out/r06_g4_kernels.txt 2>&1
tail -30 gpurun_out/r06_g4_kernels.txt
timeout 900 python tools/hevc_bench.py > gpurun_out/r06_hevc_bench.json 2> gpurun_out/r06_hevc_bench.err
cat gpurun_out/r06_hevc_bench.json; tail -3 gpurun_out/r06_hevc_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_hevc -o hevc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/hevc_bench.py --steps 2 > $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_hevc -name "*kernel_stats.csv" | head -3
f=$(find gpurun_out/prof_hevc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_hevc_kernel_stats.csv && head -20 gpurun_out/r06_hevc_kernel_stats.csv
rm -rf gpurun_out/prof_hevc
