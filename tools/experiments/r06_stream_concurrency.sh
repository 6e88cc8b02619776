/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/stream_concurrency tools/experiments/src/stream_concurrency.hip 2>/dev/null
echo default; timeout 120 /tmp/stream_concurrency 500 40
echo GPU_MAX_HW_QUEUES=16; GPU_MAX_HW_QUEUES=16 timeout 120 /tmp/stream_concurrency 500 40
echo GPU_MAX_HW_QUEUES=32; GPU_MAX_HW_QUEUES=32 timeout 120 /tmp/stream_concurrency 500 40
echo GPU_MAX_HW_QUEUES=2; GPU_MAX_HW_QUEUES=2 timeout 120 /tmp/stream_concurrency 500 40
