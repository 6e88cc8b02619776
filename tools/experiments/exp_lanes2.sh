#!/bin/bash
# lane count sweep on the bench workload, interleaved twice (boxes differ; only one box's runs compare)
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-output > /dev/null 2>&1
for rep in 1 2; do
for l in ${LANE_SET:-32 40 48 56 64 80}; do
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-output --lanes $l 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lanes', $l, d['value'], 'fps  cpu ms/pic', d['host_cpu_ms_per_picture'], 'busy', d['host_cpus_busy_rank0'])"
done
done
