# the H.264 headline with this round's library and with the library of round 5's last tree, interleaved on one box:
# did anything of round 6 (retriever: item planning, feed_would_wait; executor: linger defaults) cost the H.264 path?
for rep in 1 2; do
for lib in new old; do
if [ $lib = old ]; then export HWANG_AMD_LIB=$GRAFT_REPO_ROOT/_ab/lib_round5_end.so HWANG_AMD_LIB_ALLOW_OLD=1; else unset HWANG_AMD_LIB HWANG_AMD_LIB_ALLOW_OLD; fi
timeout 900 python bench.py --no-cpu-baseline --no-hbm-output --no-reference-api --no-hevc --steps 3 > gpurun_out/r06_ab_$lib.json 2> gpurun_out/r06_ab_$lib.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_ab_$lib.json").read().strip().splitlines()[-1])
    print("$lib:", d["value"], d["roofline"]["frac"], d.get("host_cpu_ms_per_picture"), d["config"].get("lane_threads"))
except Exception as e:
    print("$lib failed:", e, open("gpurun_out/r06_ab_$lib.err").read()[-600:])
PY
done; done
