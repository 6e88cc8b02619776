for cfg in "16 16" "8 16" "8 4" "16 64" "32 64" "4 16" "12 32"; do
set -- $cfg
HWANG_HEVC_FLOW_WORKERS=$1 HWANG_HEVC_SPIN_SLEEP=$2 timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes 16 2> gpurun_out/r06_hevc_i.err > gpurun_out/r06_hevc_i.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_i.json"))
print("workers $1 sleep $2:", d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_wall_ms_per_picture"])
PY
done
