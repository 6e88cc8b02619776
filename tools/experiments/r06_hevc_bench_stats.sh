timeout 900 python -m pytest tests/test_gpu_hevc_kernels.py tests/test_gpu_hevc.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes 24 2> gpurun_out/r06_hevc_j.err | cut -c300-640
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_hevc -o hevc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/hevc_bench.py --frames 36000 --steps 2 --lanes 24 > $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_bench11_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_prof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_hevc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_hevc_kernel_stats11.csv && head -9 gpurun_out/r06_hevc_kernel_stats11.csv | cut -c1-150
rm -rf gpurun_out/prof_hevc
