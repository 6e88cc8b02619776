# kernel times and PMC traffic of the HEVC path after the fences went (32 lanes)
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_hevc -o hevc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/hevc_bench.py --frames 36000 --steps 2 --lanes 32 > $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_bench_fence_free_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_prof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_hevc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_hevc_kernel_stats_fence_free.csv && head -10 gpurun_out/r06_hevc_kernel_stats_fence_free.csv | cut -c1-150
rm -rf gpurun_out/prof_hevc
bash tools/hevc_pmc.sh 2>&1 | tail -60
cp gpurun_out/r06_hevc_pmc_summary.json gpurun_out/r06_hevc_pmc_summary_fence_free.json
