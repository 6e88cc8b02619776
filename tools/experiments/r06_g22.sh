HWANG_HEVC_ENGINE_STATS=1 timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes 16 2> gpurun_out/r06_hevc_h.err | tee gpurun_out/r06_hevc_h.json | cut -c 300-700
tail -5 gpurun_out/r06_hevc_h.err
