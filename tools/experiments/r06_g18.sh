mkdir -p /tmp/tl
HWANG_HEVC_TIMELINE=/tmp/tl/t timeout 900 python tools/hevc_bench.py --frames 36000 --steps 1 --lanes 16 > gpurun_out/r06_hevc_tl_bench.json 2> gpurun_out/r06_hevc_tl.err
cat gpurun_out/r06_hevc_tl_bench.json; tail -3 gpurun_out/r06_hevc_tl.err
ls /tmp/tl | wc -l
python tools/experiments/hevc_timeline.py /tmp/tl/t 1.2
tar czf gpurun_out/r06_hevc_timeline.tgz -C /tmp tl
