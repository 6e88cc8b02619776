#!/bin/bash
# Development aid: the parser + probe as a small shared library of its own, compiled like the product library
# (hipcc's clang, the Makefile's flags, -fPIC), so that variants can be measured side by side (lib_counters.py).
# usage: build_variant.sh NAME [extra compiler flags]  ->  tools/hostprof/libparser_NAME.so
cd "$(dirname "$0")"
N=$1; shift
R=../..
/opt/rocm/bin/hipcc -std=c++17 -O3 -march=x86-64-v3 -fPIC -Wall -Wno-unused-function -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
  -I$R/include -I$R/hwang_amd/csrc "$@" -Wl,-Bsymbolic -shared -o libparser_$N.so \
  $R/hwang_amd/csrc/capi_probe.cpp $R/hwang_amd/csrc/h264/h264_parser.cpp $R/hwang_amd/csrc/h264/h264_cabac_init.cpp $R/hwang_amd/csrc/hevc/hvcc.cpp
