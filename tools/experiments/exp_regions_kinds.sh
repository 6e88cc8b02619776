#!/bin/bash
# Development aid (GPU box): instructions per region of the macroblock layer for keyframes alone and for GOPs that end in a B picture.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_regions_kinds.txt
CLIP=/tmp/c1_600.mp4
[ -f $CLIP ] || ./corpus/_build/h264_encoder --width 1920 --height 1080 --frames 600 --gop 60 --bframes 2 --fps 30 --qp 26 --threads 32 --out $CLIP > /dev/null
{
for w in keys b; do
  for l in v8 regions regions_fine; do
    echo "#### WANTED=$w lib=$l"
    WANTED=$w REPS=1 python3 tools/hostprof/lib_counters.py $CLIP tools/hostprof/libparser_$l.so 2>&1
  done
done
} > $OUT 2>&1
tail -5 $OUT
