#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for l in lib_head lib_chain_hoist lib_head lib_chain_hoist; do
  echo "== $l"
  HWANG_AMD_LIB=$PWD/_ab/$l.so timeout 300 python3 tools/ref_api_probe.py frames=10800 rows=108 threads=1 group_mb=0 2>&1 | tail -2
done
