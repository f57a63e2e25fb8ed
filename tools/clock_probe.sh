#!/bin/bash
# Shader clock a kernel sustains: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) and the kernel-trace duration of
# the same dispatches in one run.  clock_probe.sh <kernel substring> <program.py> [args...]
set -e -o pipefail
K=$1; shift
OUT=$PWD/gpurun_out/clock_probe
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/p" -o p -- python3 "$@" > "$OUT/log.txt" 2>&1
python3 - "$OUT" "$K" <<'PY'
import csv, glob, sys
out, k = sys.argv[1], sys.argv[2]
cyc = {}
for r in csv.DictReader(open(glob.glob(out + '/p/**/*counter_collection.csv', recursive=True)[0])):
    if k in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        cyc[r['Dispatch_Id']] = float(r['Counter_Value']) / 8.0
dur = {}
for r in csv.DictReader(open(glob.glob(out + '/p/**/*kernel_trace.csv', recursive=True)[0])):
    if k in r['Kernel_Name']:
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
for d in sorted(cyc, key=int):
    if d in dur:
        print('dispatch %s: %.3f ms, %.3e cycles -> %.3f GHz' % (d, dur[d] * 1e3, cyc[d], cyc[d] / dur[d] / 1e9))
PY
grep median "$OUT/log.txt" || true
