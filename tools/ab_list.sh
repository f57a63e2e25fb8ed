#!/bin/bash
# times every library build under tools/abrun/ (and the in-tree one) with tools/kbench.py on the same box
for l in bayesian-inference_amd/lib/libbvi_hip.so tools/abrun/*.so; do
  echo -n "$l: "; timeout -k 10 120 python tools/kbench.py --f16 --reps 20 --lib $l "$@" 2>&1 | grep -o "median [0-9.]* ms  min [0-9.]*" || echo failed
done
