# Development aid: A/B of environment settings on one box, alternating: the bench line under each.
# usage: exp_ab_env.sh tag VAR value_a value_b ...
TAG=$1; VAR=$2; shift 2
for i in 1 2 3; do
  for v in "$@"; do
    env $VAR=$v timeout 900 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-output --no-reference-api 2>/dev/null | tail -1 > gpurun_out/${TAG}_${v}_$i.json
    python3 - gpurun_out/${TAG}_${v}_$i.json "$VAR=$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
h = d.get("host_lane_threads") or {}
r = d.get("roofline") or {}
print(sys.argv[2], "value", d.get("value"), "pics/s", d.get("pictures_reconstructed_per_s"), "cpu ms/pic", d.get("host_cpu_ms_per_picture"),
      "lane cycles/pic", h.get("cycles_per_picture"), "ipc", h.get("ipc"), "frac", r.get("frac"), "launches", r.get("launches"), "parity", (d.get("parity_check") or {}).get("mismatches"))
PY
  done
done
