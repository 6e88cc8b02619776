#!/bin/bash
# first GPU bring-up: simplest clips first
set -x
E=./corpus/_build/h264_encoder
mkdir -p /tmp/gc
$E --width 320 --height 240 --frames 1 --gop 1 --bframes 0 --cabac 1 --t8x8 0 --refs 1 --out /tmp/gc/i_only.mp4
$E --width 320 --height 240 --frames 6 --gop 6 --bframes 0 --cabac 1 --t8x8 0 --refs 1 --out /tmp/gc/ip.mp4
$E --width 320 --height 240 --frames 6 --gop 6 --bframes 0 --cabac 1 --t8x8 1 --refs 2 --out /tmp/gc/ip_t8.mp4
$E --width 320 --height 240 --frames 7 --gop 7 --bframes 2 --cabac 1 --t8x8 1 --refs 2 --out /tmp/gc/ibp.mp4
for f in i_only ip ip_t8 ibp; do timeout 120 python tools/gpu_compare.py /tmp/gc/$f.mp4 2>&1 | tail -8; done
