python -c "
import os
print('affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max','/sys/fs/cgroup/cpu/cpu.cfs_quota_us','/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
"
for l in 16 24 32 48; do
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes $l 2> gpurun_out/r06_hevc_i.err > gpurun_out/r06_hevc_i.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_i.json"))
print("lanes $l:", d["value"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_wall_ms_per_picture"])
PY
done
