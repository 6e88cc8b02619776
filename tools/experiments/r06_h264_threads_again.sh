# H.264 headline: lane threads beyond the CPU quota again (the HEVC path gets 18 CPUs out of a quota of 16 with 32 threads)
for cfg in "0 0" "32 1" "48 2" "40 2" "48 1"; do
set -- $cfg
if [ "$2" = "0" ]; then unset HWANG_LANES_PER_THREAD; else export HWANG_LANES_PER_THREAD=$2; fi
timeout 900 python bench.py --no-cpu-baseline --no-hbm-output --no-reference-api --no-hevc --lanes $1 --steps 3 > gpurun_out/r06_h264_t.json 2> gpurun_out/r06_h264_t.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r06_h264_t.json").read().strip().splitlines()[-1])
print("lanes $1, lanes per thread $2:", d["value"], d["config"].get("lanes"), d["config"].get("lane_threads"), d["roofline"]["frac"], d.get("host_parse_ms_per_picture"), d.get("ms_per_step"))
PY
done
