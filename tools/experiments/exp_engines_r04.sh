#!/bin/bash
# Round 4, after the row kernels got faster: light engines x batches in flight vs bench value and roofline.frac.
cd "$GRAFT_REPO_ROOT" || exit 1
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-output --no-reference-api > /dev/null 2>&1   # corpus
run() {
  echo "== $*"
  env "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-output --no-reference-api 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=d['kernels']
print('value %.1f  frac %.4f  deblock %d launches of %.0f us  mv %d x %.0f us  intra %d x %.0f us  cpu_ms/pic %.3f' % (d['value'], r['frac'], k['deblock']['launches'], k['deblock']['avg_us'], k['mv_derivation']['launches'], k['mv_derivation']['avg_us'], k['intra_pred']['launches'], k['intra_pred']['avg_us'], d['host_cpu_ms_per_picture']))"
}
run A=1
run HWANG_LIGHT_ENGINES=2
run HWANG_LIGHT_ENGINES=1
run HWANG_LIGHT_ENGINES=2 HWANG_LIGHT_INFLIGHT=1
run HWANG_LIGHT_INFLIGHT=1
run HWANG_HEAVY_ENGINES=2
run HWANG_HEAVY_ENGINES=1
run A=1
