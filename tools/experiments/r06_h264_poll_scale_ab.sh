# poll-interval A/B on the H.264 path (bench.py, extras off): two rounds, interleaved
for round in 1 2; do for k in 1 8 32; do
HWANG_AMD_LIB=$PWD/_ab/lib_poll$k.so timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-output --no-reference-api > gpurun_out/r06_poll_$k.json 2> gpurun_out/r06_poll_$k.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_poll_$k.json").read().strip().splitlines()[-1])
    print("poll scale $k round $round:", d["value"], d["ms_per_step"], d.get("host_cpu_ms_per_picture"), d.get("pictures_reconstructed_per_s"))
except Exception as e:
    print("poll $k failed", e); print(open("gpurun_out/r06_poll_$k.err").read()[-800:])
PY
done; done
