#!/bin/bash
# Development aid (GPU box): the regions of the macroblock layer with and without one B picture per GOP - the difference is the B pictures'.
cd "$(dirname "$0")/../.."
OUT=$1
CLIP=/tmp/c1_600.mp4
[ -f $CLIP ] || ./corpus/_build/h264_encoder --width 1920 --height 1080 --frames 600 --gop 60 --bframes 2 --fps 30 --qp 26 --threads 32 --out $CLIP > /dev/null
{
for w in p b; do
    echo "#### WANTED=$w"
    WANTED=$w REPS=1 python3 tools/hostprof/lib_counters.py $CLIP tools/hostprof/libparser_regions.so 2>&1
done
} > $OUT 2>&1
