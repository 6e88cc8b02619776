timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_dev_2ranks_1gpu.json 2> gpurun_out/r06_bench_dev_2ranks_1gpu.err
echo "rc=$?"; tail -c 600 gpurun_out/r06_bench_dev_2ranks_1gpu.json | head -c 400; echo
HWANG_BENCH_DEVICE_IDS=0,0 timeout 1500 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_dev_2slots_1gpu.json 2> gpurun_out/r06_bench_dev_2slots_1gpu.err
echo "rc=$?"
python - <<PY
import json
for f in ("gpurun_out/r06_bench_dev_2ranks_1gpu.json","gpurun_out/r06_bench_dev_2slots_1gpu.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["n_gpus"], d["config"]["launch"])
    except Exception as e:
        print(f, "failed", e); print(open(f.replace(".json",".err")).read()[-800:])
PY
