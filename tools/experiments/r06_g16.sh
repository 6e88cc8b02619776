timeout 900 python -m pytest tests/test_gpu_hevc_kernels.py tests/test_gpu_hevc.py tests/test_gpu_ref_automata.py -q -m gpu > gpurun_out/r06_g16_hevc.txt 2>&1
tail -5 gpurun_out/r06_g16_hevc.txt
for q in 4 8; do for l in 16 24; do
GPU_MAX_HW_QUEUES=$q timeout 900 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes $l > gpurun_out/r06_hevc_b_q${q}_l$l.json 2> gpurun_out/r06_hevc_b_q${q}_l$l.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_b_q${q}_l$l.json"))
print("hwq", $q, "lanes", $l, d["value"], d["seconds_per_call"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["staging_stall_ms_per_picture"])
PY
done; done
