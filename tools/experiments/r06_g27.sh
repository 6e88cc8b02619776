timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r06_gpu_tests_engine.txt 2>&1
tail -5 gpurun_out/r06_gpu_tests_engine.txt
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes 24 2> gpurun_out/r06_hevc_bench9_l24.err | tee gpurun_out/r06_hevc_bench9_l24.json
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes 16 2> gpurun_out/r06_hevc_bench9_l16.err | tee gpurun_out/r06_hevc_bench9_l16.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_hevc -o hevc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/hevc_bench.py --frames 36000 --steps 2 --lanes 24 > $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_bench9_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_prof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_hevc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_hevc_kernel_stats9.csv && head -12 gpurun_out/r06_hevc_kernel_stats9.csv
rm -rf gpurun_out/prof_hevc
