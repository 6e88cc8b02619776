# Development aid: bench value and intra span vs heavy-engine configuration
for cfg in "3 3" "1 2" "1 1" "2 2" "3 3" "1 2"; do
  set -- $cfg
  echo -n "heavy engines $1 inflight $2: "
  HWANG_HEAVY_ENGINES=$1 HWANG_HEAVY_INFLIGHT=$2 timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']['intra_pred']
print('value', d['value'], 'intra launches', k['launches'], 'avg_us', k['avg_us'], 'alg GB/s', k['alg_GBps'], 'frac', d['roofline']['frac'], d['roofline']['kernel'])"
done
