set -x
timeout 1500 python -m pytest tests/test_gpu_hevc.py tests/test_gpu_ref_automata.py -x -q > gpurun_out/r06_g2_pytest.txt 2>&1
tail -30 gpurun_out/r06_g2_pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_g2_smoke.txt 2>&1
tail -5 gpurun_out/r06_g2_smoke.txt
