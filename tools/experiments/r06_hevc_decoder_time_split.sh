HWANG_HEVC_ENGINE_STATS=1 timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes 24 2> gpurun_out/r06_hevc_l.err > gpurun_out/r06_hevc_l.json
python - <<PY
import json,re
d=json.load(open("gpurun_out/r06_hevc_l.json"))
print(d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"])
tot=[0]*6
for l in open("gpurun_out/r06_hevc_l.err"):
    m=re.search(r"pictures (\d+): parse ([\d.]+) ms, waiting for a set ([\d.]+) ms, submitting ([\d.]+) ms \(([\d.]+) ms of it handing the job to the engine\), copying frames out of bounce buffers ([\d.]+)", l)
    if m:
        for i in range(6): tot[i]+=float(m.group(i+1))
print("all decoders, whole process: pictures %d parse %.0f stall %.0f submit %.0f (engine hand-over %.0f) copy-out %.0f ms" % tuple(tot))
print([l for l in open("gpurun_out/r06_hevc_l.err") if "engine" in l][-1])
PY
