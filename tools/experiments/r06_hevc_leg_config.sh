# the bench leg's configuration (32 lanes, a thread each, mean of three timed calls) beside 24 lanes on 16 threads, twice each
for rep in 1 2; do
for cfg in "32 --thread-per-lane" "24"; do
set -- $cfg
timeout 600 python tools/hevc_bench.py --frames 36000 --steps 3 --lanes $1 $2 2> gpurun_out/r06_hevc_t.err > gpurun_out/r06_hevc_leg_$1_$rep.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06_hevc_leg_$1_$rep.json"))
print("lanes $1 $2:", d["value"], d["pictures_reconstructed"], d["pictures_reconstructed_per_s"], d["host_parse_ms_per_picture"], d["lane_cpu_ms_per_picture"], d["staging_stall_ms_per_picture"], d["lane_threads"], d["work_items"], d["seconds_per_call"], d["process_cpus_used_per_call"])
PY
done; done
