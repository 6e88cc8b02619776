#!/bin/bash
# A/B of builds of the library on one box: per-kernel spans with one lane (a picture's latency through
# each kernel, nothing concurrent) and optionally with 48 lanes (wide launches).
# usage: [PL="1 48"] ab_libs.sh libA.so libB.so ...
export FRAMES=${FRAMES:-600} REPEAT=${REPEAT:-3}
LANES=8 PROF_LANES=1 python3 tools/perf_probe.py > /dev/null 2>&1   # corpus
for pl in ${PL:-1 48}; do
  for lib in "$@"; do
    echo "== $lib  PROF_LANES=$pl"
    HWANG_AMD_LIB=$PWD/$lib LANES=48 PROF_LANES=$pl python3 tools/perf_probe.py 2>&1 | grep -E "inter_pred|intra_pred|deblock|mv_deriv"
  done
done
