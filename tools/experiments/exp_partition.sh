# Development aid: bench value and kernel spans for several CU partitions (stride: every n-th CU reserved for the heavy engines)
for P in 8 4 2 0; do
  echo "== HWANG_CU_PARTITION=$P"
  HWANG_CU_PARTITION=$P timeout 600 python3 bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'pics/s', d['pictures_reconstructed_per_s'], {k:round(v['avg_us']) for k,v in d['kernels'].items()})"
done
