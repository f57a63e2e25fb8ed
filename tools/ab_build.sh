#!/bin/bash
# builds an experimental variant of one source (SRC, default linreg_f16) into tools/abrun/libbvi_<name>.so:
#   [SRC=gmm1d] tools/ab_build.sh <name> [-DFLAG ...]
set -e
name=$1; shift
SRC=${SRC:-linreg_f16}
cd "$(dirname "$0")/../bayesian-inference_amd/csrc"
make -s >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize "$@" -c $SRC.hip -o /tmp/ab_$name.o
mkdir -p ../../tools/abrun
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abrun/libbvi_$name.so $(ls build/*.o | grep -v "build/$SRC.o") /tmp/ab_$name.o
echo built tools/abrun/libbvi_$name.so
