HEVC_BENCH_THREADS=1 timeout 600 python tools/hevc_bench.py --frames 108000 --steps 3 --lanes 24 2> gpurun_out/r06_hevc_threads2.err > gpurun_out/r06_hevc_threads2.json
tail -16 gpurun_out/r06_hevc_threads2.err
python -c "
import json;d=json.load(open('gpurun_out/r06_hevc_threads2.json'));print(d['value'],d['seconds_per_call'],d['lane_cpu_ms_per_picture'],d['pictures_reconstructed'])"
