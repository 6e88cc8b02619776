#!/bin/bash
# As ab_libs.sh, for keyframes only (I pictures: intra prediction + deblocking of intra macroblocks in isolation).
export FRAMES=${FRAMES:-600} REPEAT=${REPEAT:-3} ROWS_MODE=keyframes
LANES=8 PROF_LANES=1 python3 tools/perf_probe.py > /dev/null 2>&1   # corpus
for lib in "$@"; do
  echo "== $lib  keyframes, PROF_LANES=1"
  HWANG_AMD_LIB=$PWD/$lib LANES=48 PROF_LANES=1 python3 tools/perf_probe.py 2>&1 | grep -E "intra_pred|deblock"
done
