for cfg in "100 24000 3000 0" "100 48000 6000 0" "100 24000 3000 96" "100 48000 6000 96" "100 12000 1500 96" "100 100000 12000 128"; do
set -- $cfg
HWANG_BENCH_LANES=$4 HWANG_LINGER_FILL=$1 HWANG_LINGER_US=$2 HWANG_LINGER_IDLE_US=$3 timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-output --no-reference-api --no-hevc > gpurun_out/r06_linger.json 2> gpurun_out/r06_linger.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_linger.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("fill $1 age $2 idle $3 lanes $4:", d["value"], "cpu/pic", d.get("host_cpu_ms_per_picture"), "launches", r["launches"], "avg_us", r["avg_launch_us"], "frac", r["frac"], "lanes", d["config"]["lanes_per_gpu"])
except Exception as e:
    print("failed $cfg", e); print(open("gpurun_out/r06_linger.err").read()[-600:])
PY
done
