# every closed HEVC GOP with a wanted row is a work item of its own now (the hvcC box is no longer read as an avcC box)
timeout 900 python -m pytest tests/test_gpu_hevc.py -x -q -m gpu 2>&1 | tail -2
for lanes in 24 32 40; do
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 4 --lanes $lanes 2> gpurun_out/r06_hevc_t.err > gpurun_out/r06_hevc_t.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_t.json"))
print("lanes $lanes:", d["value"], d["pictures_reconstructed"], d["pictures_skipped"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_threads"], d["work_items"], d["seconds_per_call"], d["process_cpus_used_per_call"])
PY
done
timeout 900 python tools/hevc_bench.py --frames 36000 --steps 8 --lanes 32 --soak 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('soak:', d['soak'], d['value'])"
