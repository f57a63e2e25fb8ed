#!/bin/bash
# cycles (GRBM_GUI_ACTIVE / 8 XCDs) and duration of the regression kernel for every build under tools/abrun/ and the in-tree one:
# separates what a change does to the cycle count from what the chip's clock management does to the time
export TMPDIR=/tmp
for l in bayesian-inference_amd/lib/libbvi_hip.so tools/abrun/*.so; do
  out=gpurun_out/abc/$(basename $l .so); rm -rf $out; mkdir -p $out
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o p -- python3 tools/kbench.py --f16 --reps 6 --lib $l > $out/log.txt 2>&1 || { echo "$l failed"; exit 1; }
  python3 - $out $l <<'PY'
import csv, glob, sys
d = sys.argv[1]
cyc = [float(r['Counter_Value']) / 8 for r in csv.DictReader(open(glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]))
       if 'linreg_f16' in r['Kernel_Name'] and 'finalize' not in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE']
dur = [float(r['End_Timestamp']) - float(r['Start_Timestamp']) for r in csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]))
       if 'linreg_f16' in r['Kernel_Name'] and 'finalize' not in r['Kernel_Name']]
cyc, dur = sorted(cyc)[len(cyc) // 2], sorted(dur)[len(dur) // 2]
print('%-44s %.3f Mcycles  %.3f ms  -> %.2f GHz' % (sys.argv[2], cyc / 1e6, dur / 1e6, cyc / dur))
PY
done
