# Development aid: A/B two builds of the library on the same box (bench value, alternating)
for i in 1 2 3; do
  for v in oldparser amd; do
    echo -n "$v: "
    HWANG_AMD_LIB=$GRAFT_REPO_ROOT/hwang_amd/libhwang_$v.so timeout 600 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'pics/s', d['pictures_reconstructed_per_s'])"
  done
done
