# linger sweep after the fill test was repaired (the share is taken of the lanes at work): fill %, age limit us, idle us
for cfg in "85 6000 700" "60 6000 700" "100 6000 700" "85 3000 700" "85 12000 700" "85 6000 300" "85 6000 1500" "50 3000 300" "100 12000 1500" "85 0 700"; do
set -- $cfg
HWANG_LINGER_FILL=$1 HWANG_LINGER_US=$2 HWANG_LINGER_IDLE_US=$3 timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-output --no-reference-api --no-hevc > gpurun_out/r06_linger.json 2> gpurun_out/r06_linger.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_linger.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("fill $1 age $2 idle $3:", d["value"], "cpu/pic", d.get("host_cpu_ms_per_picture"), "launches", r["launches"], "avg_us", r["avg_launch_us"], "frac", r["frac"])
except Exception as e:
    print("failed $cfg", e); print(open("gpurun_out/r06_linger.err").read()[-600:])
PY
done
