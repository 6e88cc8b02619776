#!/bin/bash
# Interleaved A/B on one box: the library of an earlier round (built by hand into _ab/, git-ignored) against the
# current one, bench value and CPU per picture.  usage: exp_ab_round.sh _ab/libhwang_r02.so [rounds]
OLD=${1:?old library}; N=${2:-3}
q() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('%7.1f frames/s  %.3f ms CPU/picture  %.2f CPUs' % (d['value'], d['host_cpu_ms_per_picture'], d['host_cpus_busy_rank0']))"; }
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-output --no-reference-api > /dev/null 2>&1
for i in $(seq 1 $N); do
  echo -n "old: "; HWANG_AMD_LIB=$GRAFT_REPO_ROOT/$OLD HWANG_AMD_LIB_ALLOW_OLD=1 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-output --no-reference-api 2>/dev/null | q
  echo -n "new: "; python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-output --no-reference-api 2>/dev/null | q
done
