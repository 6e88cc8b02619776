#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for l in ${LIBS:-st_mv5 st_mv5_nowait st_db0 st_db5}; do
  echo "=== $l"
  HWANG_AMD_LIB=$PWD/_ab/lib_$l.so timeout 200 python3 tools/experiments/exp_stamps.py 2>&1 | tail -${TAIL:-140}
done
