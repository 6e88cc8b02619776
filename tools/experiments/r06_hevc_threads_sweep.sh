# more lane threads than the CPU quota (16): do the threads that wait for the GPU leave CPU time the others can use?
for cfg in "24 16" "24 20" "24 24" "32 24" "32 32" "48 24"; do
set -- $cfg
HWANG_RETRIEVER_THREADS=$2 timeout 600 python tools/hevc_bench.py --frames 108000 --steps 2 --lanes $1 2> gpurun_out/r06_hevc_t.err > gpurun_out/r06_hevc_t.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_t.json"))
print("lanes $1 threads $2:", d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"])
PY
done
