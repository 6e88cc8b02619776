# soaks on the round's final tree: the H.264 row kernels (tools/soak.py) and the HEVC engine (tools/hevc_bench.py --soak)
timeout 1200 python tools/soak.py > gpurun_out/r06_final_soak_h264.txt 2>&1; tail -4 gpurun_out/r06_final_soak_h264.txt
for l in 8 24 32; do
timeout 900 python tools/hevc_bench.py --frames 36000 --steps 10 --lanes $l --soak 2> gpurun_out/r06_hevc_soak_$l.err > gpurun_out/r06_hevc_soak_$l.json
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r06_hevc_soak_$l.json")); print("HEVC lanes $l:", d["value"], d["soak"])
except Exception as e:
    print("lanes $l failed", e); print(open("gpurun_out/r06_hevc_soak_$l.err").read()[-1500:])
PY
done
timeout 900 python tools/hevc_bench.py --frames 36000 --steps 10 --lanes 32 --thread-per-lane --soak 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('HEVC lanes 32, a thread each:', d['value'], d['soak'])"
