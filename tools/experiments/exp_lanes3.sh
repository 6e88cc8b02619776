#!/bin/bash
# few lane counts, interleaved several times with longer runs (run-to-run noise on one box is +-5 %)
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-output > /dev/null 2>&1
for rep in 1 2 3 4; do
for l in ${LANE_SET:-24 32 48}; do
  python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-hbm-output --lanes $l 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lanes', $l, d['value'], 'fps  cpu ms/pic', d['host_cpu_ms_per_picture'], 'busy', d['host_cpus_busy_rank0'])"
done
done
