for l in 1 2 4 8 12 16 20; do
timeout 900 python tools/hevc_bench.py --frames 36000 --steps 2 --lanes $l > gpurun_out/r06_hevc_lanes_$l.json 2> gpurun_out/r06_hevc_lanes_$l.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_lanes_$l.json"))
print($l, d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["staging_stall_ms_per_picture"])
PY
done
nproc; python -c "import os;print(len(os.sched_getaffinity(0)))"; lscpu | head -20
