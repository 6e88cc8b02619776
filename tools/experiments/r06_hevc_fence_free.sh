# the intra kernels' hand-offs without acquire / release fences: parity first, then the bench, then a soak
timeout 1500 python -m pytest tests/test_gpu_hevc_kernels.py tests/test_gpu_hevc.py -x -q -m gpu 2>&1 | tail -3
for lanes in 24 32; do
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 4 --lanes $lanes 2> gpurun_out/r06_hevc_t.err > gpurun_out/r06_hevc_t.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_t.json"))
print("lanes $lanes:", d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_threads"], d["seconds_per_call"], d["process_cpus_used_per_call"])
PY
done
timeout 900 python tools/hevc_bench.py --frames 36000 --steps 12 --lanes 24 --soak 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); s=d.get('soak') or d
print('soak:', s if not isinstance(s, dict) else {k: s[k] for k in s if k in ('calls','distinct_digests','digest','digests')})"
