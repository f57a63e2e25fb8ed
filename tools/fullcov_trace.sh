#!/bin/bash
# Kernel trace of tools/fullcov_bench.py (rows as drawn, then cluster by cluster): per-kernel averages.
set -e -o pipefail
export TMPDIR=/tmp
ROOT=$PWD
for mode in "" "--clustered"; do
  out=$ROOT/gpurun_out/fc_trace${mode#--}; rm -rf "$out"
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out" -o fc --output-format csv -- python3 "$ROOT/tools/fullcov_bench.py" $mode > "$out.log" 2>&1)
  echo "== rows ${mode:-as drawn}"; grep -v amdgpu.ids "$out.log" | grep fullcov
  python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'fullcov' in r['Name']:
        print('   %-62s %3s calls  avg %.3f ms' % (r['Name'].replace('(anonymous namespace)::', '')[:62], r['Calls'], float(r['AverageNs']) / 1e6))
PY
done
