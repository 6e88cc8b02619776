set -x
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_g6_pytest.txt 2>&1
tail -6 gpurun_out/r06_g6_pytest.txt
for l in 16 32 48; do timeout 600 python tools/hevc_bench.py --lanes $l > gpurun_out/r06_hevc_bench_lanes$l.json 2> gpurun_out/r06_hevc_bench_lanes$l.err; cat gpurun_out/r06_hevc_bench_lanes$l.json; tail -2 gpurun_out/r06_hevc_bench_lanes$l.err; done
