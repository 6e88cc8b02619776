# Development aid: backtraces of the busiest threads of a running retrieve (who sits in ioctl?)
cd "$GRAFT_REPO_ROOT" || exit 1
FRAMES=1200 REPEAT=45 LANES=48,48,48,48 SAMPLE=1 PROF_LANES=1 python3 tools/perf_probe.py > gpurun_out/gdb_probe.log 2>&1 &
PID=$!
( sleep 300; kill $PID 2>/dev/null ) &
WD=$!
for i in $(seq 1 150); do grep -q "^frames" gpurun_out/gdb_probe.log 2>/dev/null && break; sleep 1; done
sleep 3
echo "python pid $PID"
for i in 1 2; do
  top -H -b -n 1 -p $PID 2>/dev/null | awk 'NR>7' | head -5
  TID=$(top -H -b -n 1 -p $PID 2>/dev/null | awk 'NR>7 && $12=="python3" {print $1; exit}')
  echo "busiest python3 thread: $TID"
  timeout 60 /opt/rocm/bin/rocgdb -p $PID -batch -ex "thread apply all bt 14" > gpurun_out/gdb_bt_$i.txt 2>&1
  grep -A16 "LWP $TID)" gpurun_out/gdb_bt_$i.txt | head -22
done
wait $PID
kill $WD 2>/dev/null
tail -2 gpurun_out/gdb_probe.log
