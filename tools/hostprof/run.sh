#!/bin/bash
# Development aid: SIGPROF sampling profile of the host parser alone (no GPU),
# on whatever machine this runs on.  usage: run.sh clip.mp4 [reps]
set -e
cd "$(dirname "$0")"
ROOT=../..
CLIP=${1:?clip.mp4}
python3 - "$CLIP" <<'PY'
import sys, numpy as np
sys.path.insert(0, "../..")
import hwang_amd as hw
idx = hw.index_video(sys.argv[1])
open("avcc.bin", "wb").write(bytes(idx.metadata_bytes()))
idx.sample_offsets_np().astype(np.uint64).tofile("offs.bin")
idx.sample_sizes_np().astype(np.uint64).tofile("sizes.bin")
PY
g++ -O3 -march=x86-64-v3 -g -no-pie -std=c++17 -I$ROOT/include -I$ROOT/hwang_amd/csrc probe_main.cpp $ROOT/hwang_amd/csrc/capi_probe.cpp \
    $ROOT/hwang_amd/csrc/h264/h264_parser.cpp $ROOT/hwang_amd/csrc/h264/h264_cabac_init.cpp $ROOT/hwang_amd/csrc/hevc/hvcc.cpp -o probe_prof
HWANG_PROBE_FAST=1 ./probe_prof "$CLIP" ${2:-20}
addr2line -f -C -e probe_prof $(cat samples.txt | tr '\n' ' ') 2>/dev/null | awk 'NR%2==1' | sed 's/hwang::h264::Parser::Impl:://; s/hwang::h264::(anonymous namespace):://; s/(.*//' | sort | uniq -c | sort -rn | head -30
