#!/bin/bash
# Builds _ab/lib_<name>.so: the product library with k_recon.hip compiled with extra flags (experiment macros,
# compiler options) - for ab_libs.sh.   usage: build_variant.sh name "-DEXP_SOMETHING ..."
set -e
name=$1; shift
mkdir -p _ab build/variant
make -s -j8 lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -march=x86-64-v3 -fPIC -Wall -Wno-unused-function -pthread "$@" \
  -c hwang_amd/csrc/amd/k_recon.hip -o build/variant/k_recon_$name.o
objs=$(find build/obj -name '*.o' ! -name 'k_recon.hip.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/lib_$name.so $objs build/variant/k_recon_$name.o -pthread
echo built _ab/lib_$name.so
