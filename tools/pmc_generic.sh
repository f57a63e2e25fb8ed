#!/bin/bash
# One PMC pass over any program: PMC_COUNTERS="..." PMC_KERNEL=<substring> tools/pmc_generic.sh <program.py> [args...]
# Prints the per-kernel mean of each counter (dispatches whose name contains PMC_KERNEL).
set -e -o pipefail
OUT=$PWD/gpurun_out/pmc_generic
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc ${PMC_COUNTERS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE} --output-format csv -d "$OUT/p" -o p -- python3 "$@" > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
f = glob.glob(sys.argv[1] + '/p/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if os.environ.get('PMC_KERNEL', '') in r['Kernel_Name']:
        acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s mean %.4g  max %.4g (n=%d)' % (c, sum(v) / len(v), max(v), len(v)))
PY
