#!/bin/bash
# Interleaved on one box: the library of commit dee87c7 (827 frames/s on its box) vs the final tree - does the changed-
# dwords rule of k_deblock (or anything since) cost host time?  (Boxes differ by +-6 %: only interleaved runs compare.)
cd "$GRAFT_REPO_ROOT" || exit 1
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-output --no-reference-api > /dev/null 2>&1
for i in 1 2 3 4; do
  for v in oldk amd; do
    echo -n "$v: "
    HWANG_AMD_LIB=$GRAFT_REPO_ROOT/hwang_amd/libhwang_$v.so HWANG_AMD_LIB_ALLOW_OLD=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-output --no-reference-api 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value', d['value'], 'cpu ms/pic', d['host_cpu_ms_per_picture'], 'deblock', d['kernels']['deblock']['avg_us'])"
  done
done | tee gpurun_out/exp_ab_mid.txt
