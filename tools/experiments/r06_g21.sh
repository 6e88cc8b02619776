for l in 16 24; do
timeout 900 python tools/hevc_bench.py --frames 108000 --steps 2 --lanes $l 2> gpurun_out/r06_hevc_g_$l.err | tee gpurun_out/r06_hevc_g_$l.json
done
