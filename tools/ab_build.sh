#!/bin/bash
# builds an experimental variant of linreg_f16.hip into tools/ab/libbvi_<name>.so:  tools/ab_build.sh <name> [-DFLAG ...]
set -e
name=$1; shift
cd "$(dirname "$0")/../bayesian-inference_amd/csrc"
make -s >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize "$@" -c linreg_f16.hip -o /tmp/f16_$name.o
mkdir -p ../../tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libbvi_$name.so build/api.o build/normal_iid.o build/gmm1d.o build/linreg.o build/linreg_bf16x3.o build/posterior.o build/gmm_fullcov.o /tmp/f16_$name.o build/hmc_fused.o build/nuts_batched.o
echo built tools/ab/libbvi_$name.so
