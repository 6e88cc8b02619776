# Development aid: the reference's unmodified caller (one DecoderAutomata, one interval at a time) with several builds
# of the library, alternating on one box.  The pybind module binds hwang_amd/libhwang_amd.so by path: the file is swapped.
cp hwang_amd/libhwang_amd.so /tmp/lib_keep.so
for i in 1 2; do
  for v in r04 oldparser new; do
    cp hwang_amd/libhwang_$v.so hwang_amd/libhwang_amd.so
    echo -n "$v: "
    timeout 600 python3 tools/ref_api_probe.py rows=270 threads=1 group_mb=0 2>&1 | tail -2 | tr '\n' ' '
    echo
  done
done
cp /tmp/lib_keep.so hwang_amd/libhwang_amd.so
