#!/bin/bash
# A/B timing of two builds of the library on the same box, interleaved (clocks differ from box to box and drift):
#   tools/ab_bench.sh [kbench options]      baseline = tools/abrun/libbvi_base.so, candidate = the in-tree build
for i in 1 2 3; do
  timeout -k 10 120 python tools/kbench.py --f16 --reps 30 --lib tools/abrun/libbvi_base.so "$@" 2>&1 | grep median | sed 's/^/base: /' || exit 1
  timeout -k 10 120 python tools/kbench.py --f16 --reps 30 "$@" 2>&1 | grep median | sed 's/^/cand: /' || exit 1
done
