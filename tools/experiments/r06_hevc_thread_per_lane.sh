# per-GOP work items; a thread per lane whatever the queue length
for lanes in 28 32 36 40 48; do
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 4 --lanes $lanes --thread-per-lane 2> gpurun_out/r06_hevc_t.err > gpurun_out/r06_hevc_t.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_t.json"))
print("lanes $lanes:", d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_threads"], d["work_items"], d["seconds_per_call"], d["process_cpus_used_per_call"])
PY
done
