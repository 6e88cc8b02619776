#!/bin/bash
# Development aid (GPU box): hardware counters of the parser alone, per picture kind, for every probe binary named.
# usage: tools/hostprof/box_counters.sh out.txt probe_a [probe_b ...]   (binaries under tools/hostprof/)
cd "$(dirname "$0")/../.."
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
CLIP=/tmp/c1_600.mp4
[ -f $CLIP ] || ./corpus/_build/h264_encoder --width 1920 --height 1080 --frames 600 --gop 60 --bframes 2 --fps 30 --qp 26 \
    --threads 32 --out $CLIP > /dev/null
python3 - $CLIP <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import hwang_amd as hw
idx = hw.index_video(sys.argv[1])
open("tools/hostprof/avcc.bin", "wb").write(bytes(idx.metadata_bytes()))
idx.sample_offsets_np().astype(np.uint64).tofile("tools/hostprof/offs.bin")
idx.sample_sizes_np().astype(np.uint64).tofile("tools/hostprof/sizes.bin")
PY
cd tools/hostprof
for p in "$@"; do
  echo "== $p"
  for i in 1 2; do HWANG_PROBE_TIMES=1 HWANG_PROBE_FAST=1 ./$p $CLIP 3 60 2>&1 | grep -E "probe:|ms per decoded|instructions|cycles|branch-misses"; done
done > ../../$OUT 2>&1
cat ../../$OUT
