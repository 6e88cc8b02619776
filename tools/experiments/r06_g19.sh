timeout 900 python -m pytest tests/test_gpu_hevc_kernels.py tests/test_gpu_hevc.py tests/test_gpu_ref_automata.py -x -q -m gpu > gpurun_out/r06_g19_hevc.txt 2>&1
tail -5 gpurun_out/r06_g19_hevc.txt
for l in 4 16 24 32; do
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes $l > gpurun_out/r06_hevc_e_l$l.json 2> gpurun_out/r06_hevc_e_l$l.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r06_hevc_e_l$l.json"))
    print("lanes", $l, d["value"], d["seconds_per_call"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["staging_stall_ms_per_picture"])
except Exception as e:
    print("lanes", $l, "failed", e); print(open("gpurun_out/r06_hevc_e_l$l.err").read()[-1500:])
PY
done
