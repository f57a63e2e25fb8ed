#!/bin/bash
# Kernel time of the regression likelihood at N = 10M, D = 64 over the number of chains: the f16-plane kernel
# (what AUTO runs at this size) for every count, the exact-f32 kernel next to it where it is still competitive.
set -e -o pipefail
for c in 1 4 16 32 64; do timeout -k 10 120 python3 -u tools/kbench.py --chains $c --prec f32; done
for c in 1 4 16 32 64 128 256 512 1024 2048; do timeout -k 10 120 python3 -u tools/kbench.py --chains $c --f16; done
