#!/bin/bash
# Drop-in path (tools/ref_api_probe.py, num_devices 16): light engines dealt to the decoder's lanes x the HIP
# runtime's hardware-queue limit (GPU_MAX_HW_QUEUES, 4 by default: more streams than that share queues).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/ref_api_probe.py rows=200 threads=16 > /dev/null 2>&1   # corpus + warm caches
for round in 1 2; do
for q in 4 16; do
for e in 4 8; do
  echo -n "GPU_MAX_HW_QUEUES=$q HWANG_LIGHT_ENGINES=$e: "
  GPU_MAX_HW_QUEUES=$q HWANG_LIGHT_ENGINES=$e python tools/ref_api_probe.py rows=1080 threads=16 group_mb=128 2>&1 | grep "frames/s"
done; done; done | tee gpurun_out/exp_ref_api_queues.txt
