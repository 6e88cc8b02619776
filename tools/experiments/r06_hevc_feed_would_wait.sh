# a thread that drives several decoders passes over one whose next picture would start by waiting for the GPU
for lanes in 24 28 32 36; do
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 4 --lanes $lanes 2> gpurun_out/r06_hevc_t.err > gpurun_out/r06_hevc_t.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_t.json"))
print("lanes $lanes:", d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_threads"], d["work_items"], d["seconds_per_call"])
PY
done
