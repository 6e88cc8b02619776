for q in 4 8 16; do for l in 8 16 24; do
GPU_MAX_HW_QUEUES=$q timeout 900 python tools/hevc_bench.py --frames 36000 --steps 2 --lanes $l > gpurun_out/r06_hevc_q${q}_l$l.json 2> gpurun_out/r06_hevc_q${q}_l$l.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_q${q}_l$l.json"))
print("hwq", $q, "lanes", $l, d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["staging_stall_ms_per_picture"])
PY
done; done
