# Development aid: throughput and process CPU under HIP/HSA runtime knobs (who burns CPU in ioctl?)
run() { echo "== $*"; env "$@" FRAMES=1200 REPEAT=45 LANES=48 PROF_LANES=1 timeout 300 python3 tools/perf_probe.py 2>&1 | grep "^lanes"; }
run A=1
run HSA_ENABLE_INTERRUPT=0
run AMD_DIRECT_DISPATCH=0
run ROC_ACTIVE_WAIT_TIMEOUT=100
run ROC_CPU_WAIT_FOR_SIGNAL=0
run A=1
