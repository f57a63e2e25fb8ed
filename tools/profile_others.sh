#!/bin/bash
# rocprofv3 kernel-trace summaries of the other workloads (small-data fused samplers, VI steps, mixtures).
set -e -o pipefail
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_others
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
run() {   # name, program...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o t -- python3 "$@" > "$OUT/$name.log" 2>&1
  local f; f=$(find "$OUT/$name" -name '*_kernel_stats.csv' | head -1)
  head -12 "$f" > "$OUT/${TAG}_${name}_kernel_stats.csv"
  grep -E "steps in|chains x|ELBO|rows=|fullcov N|ms/transition" "$OUT/$name.log" | head -12 > "$OUT/${TAG}_${name}_output.txt" || true
  rm -rf "$OUT/$name"
  echo "== $name"; cut -c1-150 "$OUT/${TAG}_${name}_kernel_stats.csv" | head -6
}
run config0 tools/config0_bench.py
run vi_cfg5 tools/vi_bench.py
run mixture tools/gmm_bench.py
run fullcov tools/fullcov_bench.py
run config3_nuts tools/config3_nuts_bench.py
ls "$OUT"
