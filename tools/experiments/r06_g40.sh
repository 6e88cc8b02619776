timeout 1500 python -m pytest tests/test_gpu_hevc.py tests/test_gpu_hevc_kernels.py tests/test_gpu_ref_automata.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes 24 2> gpurun_out/r06_hevc_j.err | cut -c300-700
