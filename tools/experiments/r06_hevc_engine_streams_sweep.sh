# the engine's streams: light streams x in flight, heavy streams x in flight (32 lanes, after the fences went)
for cfg in "2 2 2 2" "2 2 1 2" "2 2 3 2" "2 2 2 1" "2 2 2 3" "1 2 2 2" "2 3 2 2" "2 1 2 2" "2 2 4 2"; do
set -- $cfg
HWANG_HEVC_LIGHT_STREAMS=$1 HWANG_HEVC_LIGHT_INFLIGHT=$2 HWANG_HEVC_HEAVY_STREAMS=$3 HWANG_HEVC_HEAVY_INFLIGHT=$4 timeout 600 python tools/hevc_bench.py --frames 36000 --steps 4 --lanes 32 2> gpurun_out/r06_hevc_t.err > gpurun_out/r06_hevc_t.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_t.json"))
print("light $1 x $2, heavy $3 x $4:", d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["staging_stall_ms_per_picture"])
PY
done
