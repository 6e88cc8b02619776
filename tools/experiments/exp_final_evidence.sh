# GPU box: the final tree's evidence after the threading / linger change: -m gpu suite, bench line + the same command
# under rocprofv3, the driver's command line, C2 / C4.
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1800 python3 -u -m pytest tests -x -q -m gpu --timeout 600 > gpurun_out/gputest.txt 2>&1; tail -3 gpurun_out/gputest.txt
bash tools/bench_profile.sh > gpurun_out/bench_profile.log 2>&1; head -c 300 gpurun_out/bench.json; echo
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.json 2> gpurun_out/bench_driver_cmd.err; head -c 300 gpurun_out/bench_driver_cmd.json; echo
for c in C2 C4; do
  python3 bench.py --config $c --no-reference-api > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
  head -c 300 gpurun_out/bench_$c.json; echo
done
