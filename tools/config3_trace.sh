#!/bin/bash
# Kernel trace of tools/config3_nuts_bench.py: the kernels by share of the GPU time.
set -e -o pipefail
export TMPDIR=/tmp
ROOT=$PWD
out=$ROOT/gpurun_out/c3_trace; rm -rf "$out"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out" -o c3 --output-format csv -- python3 "$ROOT/tools/config3_nuts_bench.py" > "$out.log" 2>&1)
grep -v "amdgpu.ids\|^W2026\|^E2026" "$out.log" | grep "N=" || true
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print('   %-66s %5s calls  avg %7.1f us  %5.1f %%' % (r['Name'].replace('(anonymous namespace)::', '')[:66], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
