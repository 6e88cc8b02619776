for cs in 1 0 1 0; do
HWANG_HEVC_COPY_STREAM=$cs timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes 24 2> gpurun_out/r06_hevc_i.err > gpurun_out/r06_hevc_i.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_i.json"))
print("copy stream $cs:", d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_wall_ms_per_picture"])
PY
done
