cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_decode.py tests/test_gpu_configs.py tests/test_gpu_workers.py -x -q -m gpu > gpurun_out/r05_gputest_linger2.log 2>&1; tail -3 gpurun_out/r05_gputest_linger2.log
OLD="HWANG_LINGER_US=0 HWANG_HEAVY_LINGER_US=0 HWANG_PRE_LINGER_US=0 HWANG_LANES_PER_THREAD=1"
show() { python3 - $1 "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline") or {}
u = d.get("value_reference_api_unmodified") or {}
g = d.get("value_reference_api") or {}
print(sys.argv[2], "value", d.get("value"), "hbm", d.get("value_hbm_resident"), "cpu ms/pic", d.get("host_cpu_ms_per_picture"), "frac", r.get("frac"), r.get("kernel"), "launches", r.get("launches"),
      "unmod", (u.get("num_devices_1") or {}).get("value"), "grouped", g.get("value"), "parity", (d.get("parity_check") or {}).get("mismatches"))
PY
}
for i in 1 2; do
  python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r05_fin_new_$i.json 2>/dev/null; show gpurun_out/r05_fin_new_$i.json new
  env $OLD python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r05_fin_old_$i.json 2>/dev/null; show gpurun_out/r05_fin_old_$i.json old
done
for c in C2 C4; do
  python3 bench.py --config $c --no-reference-api --no-cpu-baseline > gpurun_out/r05_fin_new_$c.json 2>/dev/null; show gpurun_out/r05_fin_new_$c.json "new $c"
  env $OLD python3 bench.py --config $c --no-reference-api --no-cpu-baseline > gpurun_out/r05_fin_old_$c.json 2>/dev/null; show gpurun_out/r05_fin_old_$c.json "old $c"
done
echo "small request (240 rows), new then old:"
LANES=48 FRAMES=1200 REPEAT=20 HOST_OUT=1 python3 tools/perf_probe.py 2>&1 | grep "lanes "
env $OLD LANES=48 FRAMES=1200 REPEAT=20 HOST_OUT=1 python3 tools/perf_probe.py 2>&1 | grep "lanes "
