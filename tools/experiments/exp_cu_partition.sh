#!/bin/bash
# the heavy engine's share of the CUs (HWANG_CU_PARTITION: every n-th CU; 0 = no masks), interleaved repeats
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-output > /dev/null 2>&1
for rep in 1 2; do
for p in ${PART_SET:-8 16 0}; do
  HWANG_CU_PARTITION=$p python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-output 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('partition', $p, d['value'], 'fps | inter %.0f us %.0f GB/s | intra %.0f us | deblock %.0f us %.1f GB/s frac %.4f | mv %.0f us' % (k['inter_pred']['avg_us'], k['inter_pred']['alg_GBps'], k['intra_pred']['avg_us'], k['deblock']['avg_us'], k['deblock']['alg_GBps'], d['roofline']['frac'], k['mv_derivation']['avg_us']))"
done
done
