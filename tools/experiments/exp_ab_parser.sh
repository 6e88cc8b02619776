# Development aid: A/B of two builds of the library on one box, alternating: the bench line of each (tools/experiments/README.md)
for i in 1 2 3; do
  for v in oldparser amd; do
    HWANG_AMD_LIB=$GRAFT_REPO_ROOT/hwang_amd/libhwang_$v.so timeout 900 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-output 2>/dev/null | tail -1 > gpurun_out/r05_ab_${v}_$i.json
    python3 - gpurun_out/r05_ab_${v}_$i.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
u = d.get("value_reference_api_unmodified") or {}
g = d.get("value_reference_api") or {}
print(sys.argv[2], "value", d.get("value"), "pics/s", d.get("pictures_reconstructed_per_s"), "cpu ms/pic", d.get("host_cpu_ms_per_picture"),
      "unmodified", (u.get("num_devices_1") or {}).get("value"), "grouped", g.get("value"), "parity", (d.get("parity_check") or {}).get("mismatches"))
PY
  done
done
