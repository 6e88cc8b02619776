# the HEVC path's frames must not depend on how the engine's batches fall: many calls, several lane counts, one digest
for l in 6 16 24 40; do
timeout 900 python tools/hevc_bench.py --frames 36000 --steps 12 --lanes $l --soak 2> gpurun_out/r06_hevc_soak_$l.err > gpurun_out/r06_hevc_soak_$l.json
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r06_hevc_soak_$l.json")); print("lanes $l:", d["value"], d["soak"])
except Exception as e:
    print("lanes $l failed", e); print(open("gpurun_out/r06_hevc_soak_$l.err").read()[-1500:])
PY
done
