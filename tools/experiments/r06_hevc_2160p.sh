# HEVC Main 3840x2160 (the shape of BASELINE C2: 4K, 1 % sparse), 4 unique GOPs; sets fall back to three per decoder above 1080p
for lanes in 16 24 32; do
timeout 1500 python tools/hevc_bench.py --width 3840 --height 2160 --gops 4 --frames 18000 --steps 3 --lanes $lanes --cpu-baseline 2> gpurun_out/r06_hevc_4k.err > gpurun_out/r06_hevc_4k_l$lanes.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_4k_l$lanes.json"))
print("2160p lanes $lanes:", d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_threads"], d["work_items"], d["seconds_per_call"], d["process_cpus_used_per_call"], d["upload_MB_per_picture"], d["bytes_per_sample"], (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("parity_check"))
PY
done
