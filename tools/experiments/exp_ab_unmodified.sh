# Development aid: the reference's unmodified caller (one DecoderAutomata, one interval at a time) with several builds
# of the library, alternating on one box.  The pybind module binds hwang_amd/libhwang_amd.so by path: the file is swapped.
# usage: exp_ab_unmodified.sh name_a name_b ...   (hwang_amd/libhwang_<name>.so)
cp hwang_amd/libhwang_amd.so /tmp/lib_keep.so
for i in 1 2 3; do
  for v in "$@"; do
    cp hwang_amd/libhwang_$v.so hwang_amd/libhwang_amd.so
    echo -n "$v: "
    timeout 600 python3 tools/ref_api_probe.py rows=540 threads=1 group_mb=0 2>&1 | tail -1 | tr '\n' ' '
    timeout 600 python3 tools/ref_api_probe.py rows=1080 threads=16 group_mb=128 2>&1 | tail -1 | tr '\n' ' '
    echo
  done
done
cp /tmp/lib_keep.so hwang_amd/libhwang_amd.so
