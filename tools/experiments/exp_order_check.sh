cd $GRAFT_REPO_ROOT
OLD="HWANG_LINGER_US=0 HWANG_HEAVY_LINGER_US=0 HWANG_PRE_LINGER_US=0 HWANG_LANES_PER_THREAD=1"
echo "small request (240 rows): old, new, old, new"
for i in 1 2; do
env $OLD LANES=48 FRAMES=1200 REPEAT=20 HOST_OUT=1 python3 tools/perf_probe.py 2>&1 | grep "lanes "
LANES=48 FRAMES=1200 REPEAT=20 HOST_OUT=1 python3 tools/perf_probe.py 2>&1 | grep "lanes "
done
for c in C4 C2; do
  for i in 1 2; do
  env $OLD python3 bench.py --config $c --no-reference-api --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old $c', d['value'], d['host_cpu_ms_per_picture'])"
  python3 bench.py --config $c --no-reference-api --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new $c', d['value'], d['host_cpu_ms_per_picture'])"
  done
done
