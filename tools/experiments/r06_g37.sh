timeout 900 python -m pytest tests/test_gpu_hevc.py tests/test_gpu_hevc_kernels.py tests/test_gpu_ref_automata.py -x -q -m gpu 2>&1 | tail -2
bash tools/experiments/r06_g36.sh
for l in 16 32; do
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes $l 2> gpurun_out/r06_hevc_j.err > gpurun_out/r06_hevc_m_$l.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_m_$l.json"))
print("lanes $l:", d["value"], d["seconds_per_call"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_wall_ms_per_picture"])
PY
done
