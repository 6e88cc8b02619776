cd /root/repo
for v in amd w8 w4; do
  for mode in gopend keyframes; do
    echo "=== lib $v mode $mode"
    HWANG_AMD_LIB=/root/repo/hwang_amd/libhwang_$v.so ROWS_MODE=$mode FRAMES=240 REPEAT=2 LANES=1 PROF_LANES=1 timeout 300 python3 tools/perf_probe.py 2>&1 | grep -v "^corpus\|^frames" 
  done
done
