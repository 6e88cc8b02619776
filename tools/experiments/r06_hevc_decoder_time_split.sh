HWANG_HEVC_ENGINE_STATS=1 timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes 24 2> gpurun_out/r06_hevc_l.err > gpurun_out/r06_hevc_l.json
python - <<PY
import json,re
d=json.load(open("gpurun_out/r06_hevc_l.json"))
print(d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"])
tot=[0]*7
for l in open("gpurun_out/r06_hevc_l.err"):
    m=re.search(r"pictures (\d+): parse ([\d.]+) ms, waiting for a set ([\d.]+) ms \(([\d.]+) ms of it until the engine had launched the set's picture\), submitting ([\d.]+) ms \(([\d.]+) ms of it handing the job to the engine\), copying frames out of bounce buffers ([\d.]+)", l)
    if m:
        for i in range(7): tot[i]+=float(m.group(i+1))
print("all decoders: pictures %d (whole process); last call: parse %.0f, waiting for a set %.0f (until launched %.0f) ms; whole process: submit %.0f (hand-over %.0f), copy-out %.0f ms" % tuple(tot))
print([l for l in open("gpurun_out/r06_hevc_l.err") if "engine" in l][-1]); print([l for l in open("gpurun_out/r06_hevc_l.err") if "event waits" in l][:3])
PY
python - <<PY
import re
n=[0,0]; a=[0.0,0.0]
for l in open("gpurun_out/r06_hevc_l.err"):
    m=re.search(r"waited for a set (\d+) times behind an inter picture \(submitted ([\d.]+) ms before, on average\), (\d+) times behind an intra picture \(([\d.]+) ms\)", l)
    if m:
        n[0]+=int(m.group(1)); a[0]+=int(m.group(1))*float(m.group(2)); n[1]+=int(m.group(3)); a[1]+=int(m.group(3))*float(m.group(4))
print("waits behind inter pictures: %d (age %.2f ms), behind intra pictures: %d (age %.2f ms)" % (n[0], a[0]/max(n[0],1), n[1], a[1]/max(n[1],1)))
PY
