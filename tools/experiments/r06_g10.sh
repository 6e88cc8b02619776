set -x
timeout 900 python -m pytest tests/test_gpu_hevc_kernels.py tests/test_gpu_hevc.py -x -q -m gpu > gpurun_out/r06_g10_hevc.txt 2>&1
tail -5 gpurun_out/r06_g10_hevc.txt
timeout 900 python tools/hevc_bench.py > gpurun_out/r06_hevc_bench6.json 2> gpurun_out/r06_hevc_bench6.err
cat gpurun_out/r06_hevc_bench6.json; tail -3 gpurun_out/r06_hevc_bench6.err
timeout 900 python tools/hevc_bench.py --frames 36000 --steps 2 > gpurun_out/r06_hevc_bench6_36000.json 2> gpurun_out/r06_hevc_bench6_36000.err
cat gpurun_out/r06_hevc_bench6_36000.json; tail -3 gpurun_out/r06_hevc_bench6_36000.err
timeout 900 python tools/hevc_bench.py --frames 36000 --steps 2 --lanes 24 > gpurun_out/r06_hevc_bench6_36000_l32.json 2> gpurun_out/r06_hevc_bench6_36000_l32.err
cat gpurun_out/r06_hevc_bench6_36000_l32.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_hevc -o hevc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/hevc_bench.py --frames 36000 --steps 2 > $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_bench6_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_prof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_hevc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_hevc_kernel_stats6.csv && head -12 gpurun_out/r06_hevc_kernel_stats6.csv
rm -rf gpurun_out/prof_hevc
GPU_MAX_HW_QUEUES=16 timeout 900 python tools/hevc_bench.py --frames 36000 --steps 2 > gpurun_out/r06_hevc_bench6_36000_hwq16.json 2> gpurun_out/r06_hevc_bench6_36000_hwq16.err
cat gpurun_out/r06_hevc_bench6_36000_hwq16.json
GPU_MAX_HW_QUEUES=16 timeout 900 python tools/hevc_bench.py --frames 36000 --steps 2 --lanes 32 > gpurun_out/r06_hevc_bench6_36000_hwq16_l32.json 2> gpurun_out/r06_hevc_bench6_36000_hwq16_l32.err
cat gpurun_out/r06_hevc_bench6_36000_hwq16_l32.json
