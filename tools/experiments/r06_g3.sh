set -x
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_g3_pytest.txt 2>&1
tail -15 gpurun_out/r06_g3_pytest.txt
