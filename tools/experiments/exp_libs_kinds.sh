#!/bin/bash
# Development aid (GPU box): instructions / cycles / branch misses per picture kind for the parser libraries named (tools/hostprof/libparser_NAME.so).
# usage: exp_libs_kinds.sh OUT NAME [NAME ...]
cd "$(dirname "$0")/../.."
OUT=$1; shift
CLIP=/tmp/c1_600.mp4
[ -f $CLIP ] || ./corpus/_build/h264_encoder --width 1920 --height 1080 --frames 600 --gop 60 --bframes 2 --fps 30 --qp 26 --threads 32 --out $CLIP > /dev/null
{
for rep in 1 2; do
  for l in "$@"; do
    WANTED=b REPS=2 python3 tools/hostprof/lib_counters.py $CLIP tools/hostprof/libparser_$l.so 2>&1 | tail -4
  done
done
} > $OUT 2>&1
cat $OUT
