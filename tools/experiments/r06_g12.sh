set -x
timeout 900 python -m pytest tests/test_gpu_hevc_kernels.py tests/test_gpu_hevc.py -q -m gpu > gpurun_out/r06_g12_hevc.txt 2>&1
tail -5 gpurun_out/r06_g12_hevc.txt
timeout 900 python tools/hevc_bench.py --frames 36000 --steps 2 > gpurun_out/r06_hevc_bench8_36000.json 2> gpurun_out/r06_hevc_bench8_36000.err
cat gpurun_out/r06_hevc_bench8_36000.json; tail -3 gpurun_out/r06_hevc_bench8_36000.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_hevc -o hevc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/hevc_bench.py --frames 36000 --steps 2 > $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_bench8_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_prof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_hevc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_hevc_kernel_stats8.csv && head -12 gpurun_out/r06_hevc_kernel_stats8.csv
rm -rf gpurun_out/prof_hevc
