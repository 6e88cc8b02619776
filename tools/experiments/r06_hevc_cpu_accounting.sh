# how many CPUs the whole process uses during a call (the quota is 16), and how long the cgroup froze its threads
cat /sys/fs/cgroup/cpu.max
for lanes in 24 36; do
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 4 --lanes $lanes 2> gpurun_out/r06_hevc_t.err > gpurun_out/r06_hevc_t.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_t.json"))
print("lanes $lanes:", d["value"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_threads"], d["seconds_per_call"], d["process_cpus_used_per_call"], d["cgroup_throttled_ms_per_call"])
PY
done
