timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes 16 2> gpurun_out/r06_hevc_f.err | tee gpurun_out/r06_hevc_f.json
tail -5 gpurun_out/r06_hevc_f.err
