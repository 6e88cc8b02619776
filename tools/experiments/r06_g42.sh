timeout 900 python tools/hevc_bench.py --frames 36000 --steps 2 --lanes 24 --cpu-baseline 2> gpurun_out/r06_hevc_cpu.err > gpurun_out/r06_hevc_cpu.json
tail -3 gpurun_out/r06_hevc_cpu.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_cpu.json"))
print(d["value"], json.dumps(d["cpu_baseline"]))
PY
