#!/bin/bash
# Development aid (GPU box): regions of the macroblock layer for the keyframes alone, for the region libraries named.
cd "$(dirname "$0")/../.."
OUT=$1; shift
CLIP=/tmp/c1_600.mp4
[ -f $CLIP ] || ./corpus/_build/h264_encoder --width 1920 --height 1080 --frames 600 --gop 60 --bframes 2 --fps 30 --qp 26 --threads 32 --out $CLIP > /dev/null
{
for l in "$@"; do
    echo "#### WANTED=keys lib=$l"
    WANTED=keys REPS=1 python3 tools/hostprof/lib_counters.py $CLIP tools/hostprof/libparser_$l.so 2>&1
done
} > $OUT 2>&1
