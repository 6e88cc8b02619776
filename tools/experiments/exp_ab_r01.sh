#!/bin/bash
# In-run A/B against the round-1 tree (a built copy under _ab_r01/, not tracked): boxes differ by +-5 %,
# so only interleaved runs on one box say anything.
q() { python3 -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2', d['value'], 'fps  ms/step', d['ms_per_step'], ' cpus busy', d.get('host_cpus_busy_rank0'), ' cpu ms/pic', d.get('host_cpu_ms_per_picture', round(1000*d['host_cpus_busy_rank0']/d['pictures_reconstructed_per_s'],3)))"; }
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-output > /dev/null 2>&1   # corpus + warm page cache
for i in 1 2 3; do
  (cd _ab_r01 && python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > ../gpurun_out/ab_old.json 2>/dev/null); q gpurun_out/ab_old.json "r01  "
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-output > gpurun_out/ab_new.json 2>/dev/null; q gpurun_out/ab_new.json "now  "
done
