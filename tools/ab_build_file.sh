#!/bin/bash
# Builds tools/abrun/libbvi_<name>.so from the in-tree objects with ONE source recompiled under extra flags:
#   tools/ab_build_file.sh <name> <source.hip> [-DFLAG ...]
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../bayesian-inference_amd/csrc"
mkdir -p ../../tools/abrun/b
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize "$@" \
  -c $src -o ../../tools/abrun/b/${src%.hip}_$name.o
objs=$(ls build/*.o | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abrun/libbvi_$name.so $objs ../../tools/abrun/b/${src%.hip}_$name.o
rm -f ../../tools/abrun/b/${src%.hip}_$name.o
