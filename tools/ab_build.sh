#!/bin/bash
# Builds tools/abrun/libbvi_<name>.so from the in-tree objects with linreg_f16.hip recompiled under extra flags:
#   tools/ab_build.sh <name> [-DFLAG ...]
set -e
name=$1; shift
cd "$(dirname "$0")/../bayesian-inference_amd/csrc"
mkdir -p ../../tools/abrun/b
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize "$@" \
  -c linreg_f16.hip -o ../../tools/abrun/b/linreg_f16_$name.o
objs=$(ls build/*.o | grep -v linreg_f16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abrun/libbvi_$name.so $objs ../../tools/abrun/b/linreg_f16_$name.o
rm -f ../../tools/abrun/b/linreg_f16_$name.o
