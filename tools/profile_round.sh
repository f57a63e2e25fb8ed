#!/bin/bash
# Runs on the GPU box: bench alone, then under rocprofv3 (kernel trace; FETCH_SIZE; WRITE_SIZE, each its own pass).
set -e -o pipefail
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 240 python3 bench.py > "$OUT/bench.json"
tail -1 "$OUT/bench.json"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_trace.json"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_fetch.json"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_write.json"
python3 tools/summarize_profiles.py "$TAG" "$OUT/trace" "$OUT/fetch" "$OUT/write" "$OUT/bench_trace.json"
cp profiles/${TAG}_* "$OUT/"
# keep the merge-back small: the raw traces are not needed once summarised
rm -rf "$OUT/trace" "$OUT/fetch" "$OUT/write"
ls -la "$OUT"
