#!/bin/bash
# Development aid: the host parser alone on whatever machine this runs on (the GPU box): one thread unthrottled,
# then N threads - so the bench's CPU-ms per picture can be read against what one free core does.
# usage: tools/hostprof/box_probe.sh [out.txt]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/host_box_probe.txt}
mkdir -p "$(dirname "$OUT")"
CLIP=/tmp/c1_600.mp4
[ -f $CLIP ] || ./corpus/_build/h264_encoder --width 1920 --height 1080 --frames 600 --gop 60 --bframes 2 --fps 30 --qp 26 \
    --threads 32 --out $CLIP > /dev/null
{
  lscpu | grep -E "Model name|^CPU\(s\)|Thread|MHz"
  cat /sys/fs/cgroup/cpu.max 2>/dev/null
  python3 - $CLIP <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import hwang_amd as hw
idx = hw.index_video(sys.argv[1])
open("tools/hostprof/avcc.bin", "wb").write(bytes(idx.metadata_bytes()))
idx.sample_offsets_np().astype(np.uint64).tofile("tools/hostprof/offs.bin")
idx.sample_sizes_np().astype(np.uint64).tofile("tools/hostprof/sizes.bin")
PY
  (cd tools/hostprof && for i in 1 2 3; do HWANG_PROBE_TIMES=1 HWANG_PROBE_FAST=1 ./probe_prof $CLIP 3 60; done) 2>&1
  THREADS=${THREADS:-1,8,16,24,32,48} GOPS=4 python3 tools/host_scaling.py $CLIP
} 2>&1 | tee "$OUT"
