# Development aid: bench value for several lane counts (steps 5 each, no CPU baseline)
for L in 32 40 48 64 32 40 48 64; do
  echo -n "lanes $L: "
  timeout 600 python3 bench.py --steps 5 --warmup 1 --lanes $L --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'pics/s', d['pictures_reconstructed_per_s'])"
done
