#!/bin/bash
# Kernel trace of the one-time pack step at the headline shape (bench.py, two steps): per-kernel averages of the
# f16_* kernels (bench.py packs a 4096-row slice first to load the code: the totals are the full-size launches' to
# within microseconds).
set -e -o pipefail
export TMPDIR=/tmp
ROOT=$PWD
out=$ROOT/gpurun_out/pack_trace; rm -rf "$out"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out" -o pk --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out.log" 2>&1)
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'f16_' in r['Name'] and 'linreg_f16' not in r['Name']:
        print('   %-62s %3s calls  total %.3f ms' % (r['Name'].replace('(anonymous namespace)::', '')[:62], r['Calls'], float(r['TotalDurationNs']) / 1e6))
PY
