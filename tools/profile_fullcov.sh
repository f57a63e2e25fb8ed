#!/bin/bash
# End-of-round refresh of the full-covariance records: outputs in both row orders, counters of both kernels (one
# profiler pass per run), the config-3 NUTS trace.
set -e -o pipefail
OUT=$PWD/gpurun_out/prof_fc
rm -rf "$OUT"; mkdir -p "$OUT"
{ timeout -k 10 120 python3 tools/fullcov_bench.py; timeout -k 10 120 python3 tools/fullcov_bench.py --clustered; } 2>&1 | grep -v amdgpu.ids > "$OUT/r3_fullcov_clustered_output.txt"
PMC_PROG='tools/fullcov_bench.py' PMC_KERNEL=fullcov_f16_kernel tools/pmc_kernel.sh > "$OUT/r3_pmc_fullcov.txt" 2>&1
PMC_PROG='tools/fullcov_bench.py --clustered' PMC_KERNEL=fullcov_f16_kernel tools/pmc_kernel.sh > "$OUT/r3_pmc_fullcov_clustered.txt" 2>&1
{ PMC_KERNEL=fullcov_single_kernel PMC_COUNTERS="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" tools/pmc_generic.sh tools/fullcov_bench.py --clustered;
  PMC_KERNEL=fullcov_single_kernel PMC_COUNTERS="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" tools/pmc_generic.sh tools/fullcov_bench.py --clustered; } > "$OUT/r3b_pmc_fullcov_single.txt" 2>&1
tools/fullcov_trace.sh 2>&1 | grep -v "^W2026\|^E2026\|simple_timer" > "$OUT/r3b_fullcov_trace.txt"
tools/config3_trace.sh > "$OUT/r3b_config3_nuts_trace.txt" 2>&1
ls -la "$OUT"
