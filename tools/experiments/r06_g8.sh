set -x
timeout 900 python -m pytest tests/test_gpu_hevc_kernels.py tests/test_gpu_hevc.py -x -q -m gpu > gpurun_out/r06_g9_hevc.txt 2>&1
tail -5 gpurun_out/r06_g9_hevc.txt
timeout 900 python tools/hevc_bench.py > gpurun_out/r06_hevc_bench5.json 2> gpurun_out/r06_hevc_bench5.err
cat gpurun_out/r06_hevc_bench5.json; tail -3 gpurun_out/r06_hevc_bench5.err
timeout 900 python tools/hevc_bench.py --frames 36000 --steps 2 > gpurun_out/r06_hevc_bench5_36000.json 2> gpurun_out/r06_hevc_bench5_36000.err
cat gpurun_out/r06_hevc_bench5_36000.json; tail -3 gpurun_out/r06_hevc_bench5_36000.err
timeout 900 python tools/hevc_bench.py --frames 36000 --steps 2 --lanes 24 > gpurun_out/r06_hevc_bench5_36000_l32.json 2> gpurun_out/r06_hevc_bench5_36000_l32.err
cat gpurun_out/r06_hevc_bench5_36000_l32.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_hevc -o hevc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/hevc_bench.py --frames 36000 --steps 2 > $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_bench5_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_prof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_hevc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_hevc_kernel_stats5.csv && head -12 gpurun_out/r06_hevc_kernel_stats5.csv
rm -rf gpurun_out/prof_hevc
