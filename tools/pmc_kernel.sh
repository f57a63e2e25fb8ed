#!/bin/bash
# MFMA utilisation / clock counters for one kernel (one PMC pass, SQ + GRBM blocks only).  Defaults: the headline
# regression kernel under tools/kbench.py; PMC_PROG='tools/fullcov_bench.py' PMC_KERNEL=fullcov_f16_kernel for another.
set -e -o pipefail
OUT=$PWD/gpurun_out/pmc_kernel
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p" -o p -- python3 ${PMC_PROG:-tools/kbench.py --f16 --reps 3} > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
f = glob.glob(sys.argv[1] + '/p/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if os.environ.get('PMC_KERNEL', 'linreg_f16_pipe_kernel') in r['Kernel_Name']:
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
import json
out = {}
for k, d in acc.items():
    print(k)
    out[k] = {}
    for c, v in sorted(d.items()):
        print('   %-28s %.4g  (n=%d)' % (c, sum(v) / len(v), len(v)))
        out[k][c] = sum(v) / len(v)
    m = out[k]
    if 'GRBM_GUI_ACTIVE' in m and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        cyc = m['GRBM_GUI_ACTIVE'] / 8.0                      # summed over the 8 XCDs
        m['derived_mfma_busy_fraction'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0)   # 1024 SIMDs
        print('   MFMA pipe busy %.1f %% of SIMD cycles' % (100 * m['derived_mfma_busy_fraction']))
json.dump({'command': 'rocprofv3 --pmc <SQ/GRBM counters> -- python3 tools/kbench.py --f16 --reps 3 (N=10M, D=64, 1024 chains)',
           'kernels': out}, open(sys.argv[1] + '/pmc_mfma.json', 'w'), indent=1)
PY
