#!/bin/bash
# Round-robin timing of several builds of the library on one box: tools/ab_multi.sh <rounds> <name> [<name> ...]
# (tools/abrun/libbvi_<name>.so, built by tools/ab_build.sh); prints the median / min of tools/kbench.py per build and round.
rounds=$1; shift
for i in $(seq 1 "$rounds"); do
  for nm in "$@"; do
    timeout -k 10 120 python tools/kbench.py --f16 --reps 30 --lib tools/abrun/libbvi_$nm.so ${KBENCH_ARGS:-} 2>&1 | grep median | sed "s/^/$nm: /" || exit 1
  done
done
