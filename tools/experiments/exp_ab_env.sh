# Development aid: interleaved A/B of two environment settings on the C1 probe (returned frames/s, process CPU)
# usage: exp_ab_env.sh "A_ENV=1 ..." "B_ENV=2 ..." [rounds]
A="$1"; B="$2"; N=${3:-5}
for i in $(seq 1 $N); do
  for cfg in "$A" "$B"; do
    echo -n "[$cfg] "
    env $cfg FRAMES=1200 REPEAT=45 LANES=48 PROF_LANES=1 timeout 300 python3 tools/perf_probe.py 2>&1 | grep "^lanes" | awk '{print $5, "fps  cpu", $(NF-3)}'
  done
done
