cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_hevc_trace -o hevc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/hevc_bench.py --frames 36000 --steps 3 --lanes 32 > $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_trace_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_trace.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_hevc_trace -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-22 {} | gzip > gpurun_out/r06_hevc_kernel_trace.csv.gz'
ls -la gpurun_out/r06_hevc_kernel_trace.csv.gz
rm -rf gpurun_out/prof_hevc_trace
python tools/experiments/src/hevc_trace_streams.py gpurun_out/r06_hevc_kernel_trace.csv.gz | tee gpurun_out/r06_hevc_trace_streams.txt
