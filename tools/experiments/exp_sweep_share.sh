# Development aid: linger settings, alternating on one box (bench line of each).
run() {  # tag "ENV=.. ENV=.." lanes
  env $2 timeout 900 python3 bench.py --lanes $3 --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-output --no-reference-api 2>/dev/null | tail -1 > gpurun_out/r05_sweep_$1.json
  python3 - gpurun_out/r05_sweep_$1.json "$2 lanes=$3" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline") or {}
k = d.get("kernels") or {}
def g(n, f): return (k.get(n) or {}).get(f)
print(sys.argv[2].replace("HWANG_", ""), "| value", d.get("value"), "cpu ms/pic", d.get("host_cpu_ms_per_picture"), "cpus", d.get("host_cpus_busy_rank0"),
      "| roofline", r.get("kernel"), r.get("frac"), "| deblock", g("deblock", "launches"), g("deblock", "avg_us"), "intra", g("intra_pred", "launches"), g("intra_pred", "avg_us"),
      "inter", g("inter_pred", "launches"), "mv", g("mv_derivation", "launches"), g("mv_derivation", "avg_us"))
PY
}
for i in 1 2; do
  run a$i "HWANG_LINGER_US=6000" 48
  run b$i "HWANG_LINGER_US=9000" 48
  run c$i "HWANG_LINGER_US=12000 HWANG_HEAVY_LINGER_US=9000" 48
  run d$i "HWANG_LINGER_US=9000 HWANG_HEAVY_LINGER_US=10000 HWANG_PRE_LINGER_US=3500" 48
done
