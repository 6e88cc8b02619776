# after the fences went: lanes and lane threads again (the quota is 16 CPUs)
for cfg in "24 8" "32 8" "32 4" "40 8" "48 8" "48 4" "64 8"; do
set -- $cfg
HWANG_LONG_QUEUE_FROM=$2 timeout 600 python tools/hevc_bench.py --frames 36000 --steps 4 --lanes $1 2> gpurun_out/r06_hevc_t.err > gpurun_out/r06_hevc_t.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_t.json"))
print("lanes $1 (long queue from $2 items a lane):", d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_threads"], d["seconds_per_call"], d["process_cpus_used_per_call"], d["cgroup_throttled_ms_per_call"])
PY
done
