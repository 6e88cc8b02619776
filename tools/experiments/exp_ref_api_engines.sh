for cfg in "1 1 1 64" "3 1 1 64" "1 3 1 64" "3 1 2 64" "3 2 2 256" "1 1 1 256"; do
  set -- $cfg
  echo "heavy_engines $1 heavy_inflight $2 light_inflight $3 group_mb $4"
  HWANG_HEAVY_ENGINES=$1 HWANG_HEAVY_INFLIGHT=$2 HWANG_LIGHT_INFLIGHT=$3 HWANG_EXEC_DEBUG=1 python tools/ref_api_probe.py rows=540 threads=16 group_mb=$4 > gpurun_out/probe_s.log 2>&1
  grep "num_devices\|light0\] batches\|heavy.\] batches [1-9]" gpurun_out/probe_s.log | tail -5 | cut -c1-200
done
