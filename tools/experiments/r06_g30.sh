t0=$(date +%s)
timeout 1500 python bench.py > gpurun_out/r06_bench_default_a.json 2> gpurun_out/r06_bench_default_a.err
echo "rc=$? seconds=$(( $(date +%s) - t0 ))"
python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_default_a.json").read().strip().splitlines()[-1])
print(d["value"], d["config"]["lane_threads"], d["config"]["workload"][:120])
print("hevc", json.dumps(d["hevc_main_1080p"])[:600])
print("roofline", d["roofline"])
print("cpu", d["cpu_baseline"])
print("unmodified", json.dumps(d["value_reference_api_unmodified"])[:300])
PY
tail -3 gpurun_out/r06_bench_default_a.err
