# this round's library with this round's linger defaults (100 % / 12 ms / 1.5 ms) and with round 5's numbers (85 / 6 / 0.7;
# the fill share means something different since the repair), interleaved, and round 5's library beside them
for rep in 1 2; do
for cfg in "new-defaults" "new-85-6000-700" "new-60-3000-300" "old"; do
unset HWANG_AMD_LIB HWANG_AMD_LIB_ALLOW_OLD HWANG_LINGER_FILL HWANG_LINGER_US HWANG_LINGER_IDLE_US
case $cfg in
  old) export HWANG_AMD_LIB=$GRAFT_REPO_ROOT/_ab/lib_round5_end.so HWANG_AMD_LIB_ALLOW_OLD=1;;
  new-85-6000-700) export HWANG_LINGER_FILL=85 HWANG_LINGER_US=6000 HWANG_LINGER_IDLE_US=700;;
  new-60-3000-300) export HWANG_LINGER_FILL=60 HWANG_LINGER_US=3000 HWANG_LINGER_IDLE_US=300;;
esac
timeout 900 python bench.py --no-cpu-baseline --no-hbm-output --no-reference-api --no-hevc --steps 3 > gpurun_out/r06_ab_t.json 2> gpurun_out/r06_ab_t.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_ab_t.json").read().strip().splitlines()[-1])
    print("$cfg:", d["value"], d["roofline"]["frac"], d.get("host_cpu_ms_per_picture"))
except Exception as e:
    print("$cfg failed:", e)
PY
done; done
