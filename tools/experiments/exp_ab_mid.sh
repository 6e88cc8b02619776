#!/bin/bash
# Interleaved on one box: hwang_amd/libhwang_oldk.so vs the tree's library - does the changed-dwords rule of k_deblock
# cost rate?  (Boxes differ by +-6 %: only interleaved runs compare.)  libhwang_oldk.so = this tree's host objects
# linked with an older k_recon.hip, built here before the call:
#   git show <commit>:hwang_amd/csrc/amd/k_recon.hip > build/variant/k_recon_old.hip
#   hipcc --offload-arch=gfx950 -std=c++17 -O3 -march=x86-64-v3 -fPIC -Ihwang_amd/csrc/amd -c build/variant/k_recon_old.hip -o build/variant/k_recon_old.o
#   hipcc --offload-arch=gfx950 -shared -fPIC -o hwang_amd/libhwang_oldk.so $(find build/obj -name '*.o' ! -name 'k_recon.hip.o') build/variant/k_recon_old.o -pthread
cd "$GRAFT_REPO_ROOT" || exit 1
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-output --no-reference-api > /dev/null 2>&1
for i in 1 2 3 4; do
  for v in oldk amd; do
    echo -n "$v: "
    HWANG_AMD_LIB=$GRAFT_REPO_ROOT/hwang_amd/libhwang_$v.so HWANG_AMD_LIB_ALLOW_OLD=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-output --no-reference-api 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value', d['value'], 'cpu ms/pic', d['host_cpu_ms_per_picture'], 'deblock', d['kernels']['deblock']['avg_us'])"
  done
done | tee gpurun_out/exp_ab_mid.txt
