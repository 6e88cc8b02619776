timeout 600 python -m pytest tests/test_gpu_hevc_kernels.py tests/test_gpu_hevc.py -x -q -m gpu 2>&1 | tail -2
bash tools/hevc_pmc.sh 2>&1 | tail -48 | grep -A3 "k_hevc_inter\|k_hevc_sao"
