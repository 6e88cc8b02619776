timeout 1200 python -m pytest tests/test_gpu_hevc.py tests/test_gpu_hevc_kernels.py tests/test_gpu_ref_automata.py -x -q -m gpu 2>&1 | tail -2
for l in 16 24 32; do
HWANG_HEVC_ENGINE_STATS=1 timeout 900 python tools/hevc_bench.py --frames 36000 --steps 6 --lanes $l --soak 2> gpurun_out/r06_hevc_e4_$l.err > gpurun_out/r06_hevc_e4_$l.json
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r06_hevc_e4_$l.json")); print("lanes $l:", d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["soak"])
except Exception as e:
    print("lanes $l failed", e); print(open("gpurun_out/r06_hevc_e4_$l.err").read()[-1500:])
PY
grep "hevc engine" gpurun_out/r06_hevc_e4_$l.err | tail -1
done
