echo default; timeout 120 tools/experiments/stream_concurrency 500 40
echo GPU_MAX_HW_QUEUES=16; GPU_MAX_HW_QUEUES=16 timeout 120 tools/experiments/stream_concurrency 500 40
echo GPU_MAX_HW_QUEUES=32; GPU_MAX_HW_QUEUES=32 timeout 120 tools/experiments/stream_concurrency 500 40
echo GPU_MAX_HW_QUEUES=2; GPU_MAX_HW_QUEUES=2 timeout 120 tools/experiments/stream_concurrency 500 40
