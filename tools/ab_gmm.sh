#!/bin/bash
# times the in-tree build and the named variants (tools/abrun/libbvi_<name>.so) of the mixture kernel on one box
run() { timeout -k 10 200 python tools/gmm_bench.py --only gmm1d --chains 64 1024 ${2:+--lib $2} 2>&1 | grep gmm1d | sed "s/^/$1: /" || exit 1; }
run tree
for v in "$@"; do run $v tools/abrun/libbvi_$v.so; done
