# Development aid: A/B of library builds on one box, alternating: the bench line of each (value, CPU per picture, the
# lane threads' cycles per picture).  usage: exp_ab_libs2.sh tag name_a name_b ...   (hwang_amd/libhwang_<name>.so)
TAG=$1; shift
for i in 1 2 3; do
  for v in "$@"; do
    HWANG_AMD_LIB=$GRAFT_REPO_ROOT/hwang_amd/libhwang_$v.so timeout 900 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-output --no-reference-api 2>/dev/null | tail -1 > gpurun_out/${TAG}_${v}_$i.json
    python3 - gpurun_out/${TAG}_${v}_$i.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
h = d.get("host_lane_threads") or {}
print(sys.argv[2], "value", d.get("value"), "pics/s", d.get("pictures_reconstructed_per_s"), "cpu ms/pic", d.get("host_cpu_ms_per_picture"),
      "lane cycles/pic", h.get("cycles_per_picture"), "instr/pic", h.get("instructions_per_picture"), "ghz", h.get("ghz_while_running"))
PY
  done
done
