#!/bin/bash
# rocprofv3 kernel-trace summary of one program: profile_one.sh <name> <program.py> [args...]
set -e -o pipefail
NAME=$1; shift
OUT=$PWD/gpurun_out/prof_one
rm -rf "$OUT/$NAME"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$NAME" -o t -- python3 "$@" > "$OUT/$NAME.log" 2>&1
f=$(find "$OUT/$NAME" -name '*_kernel_stats.csv' | head -1)
head -16 "$f" > "$OUT/r1_${NAME}_kernel_stats.csv"
grep -v "amdgpu.ids" "$OUT/$NAME.log" | tail -8 > "$OUT/r1_${NAME}_output.txt" || true
rm -rf "$OUT/$NAME"
cut -c1-170 "$OUT/r1_${NAME}_kernel_stats.csv"
cat "$OUT/r1_${NAME}_output.txt"
