for l in 16 24 32 48; do
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes $l 2> gpurun_out/r06_hevc_j.err > gpurun_out/r06_hevc_n_$l.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_n_$l.json"))
print("lanes $l:", d["value"], d["seconds_per_call"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_wall_ms_per_picture"])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_hevc -o hevc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/hevc_bench.py --frames 36000 --steps 2 --lanes 24 > $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_bench10_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_hevc_prof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_hevc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_hevc_kernel_stats10.csv && head -12 gpurun_out/r06_hevc_kernel_stats10.csv
rm -rf gpurun_out/prof_hevc
