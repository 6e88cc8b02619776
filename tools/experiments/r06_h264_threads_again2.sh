# H.264 headline: more lane threads than the quota's 16 CPUs WITH several lanes a thread (launch width kept)
for cfg in "0 0" "64 2" "96 3" "72 3" "60 3" "80 4"; do
set -- $cfg
if [ "$2" = "0" ]; then unset HWANG_LANES_PER_THREAD; else export HWANG_LANES_PER_THREAD=$2; fi
timeout 900 python bench.py --no-cpu-baseline --no-hbm-output --no-reference-api --no-hevc --lanes $1 --steps 3 > gpurun_out/r06_h264_t.json 2> gpurun_out/r06_h264_t.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r06_h264_t.json").read().strip().splitlines()[-1])
print("lanes $1, lanes per thread $2:", d["value"], d["config"].get("lanes_per_gpu"), d["config"].get("lane_threads"), d["roofline"]["frac"], d.get("host_cpu_ms_per_picture"), d.get("host_cpus_busy_rank0"))
PY
done
