#!/bin/bash
# Round-3 extras, run on the GPU box after tools/profile_round.sh: MFMA / issue counters of the headline and the
# full-covariance kernels (rows as drawn and cluster by cluster), the mixture kernels, the chain sweep and the
# few-chain mixture timings.  Every profiler pass is its own run under its own time limit.
set -e -o pipefail
OUT=$PWD/gpurun_out/prof_r3x
rm -rf "$OUT"; mkdir -p "$OUT"
tools/pmc_kernel.sh > "$OUT/pmc_mfma.log" 2>&1 && cp gpurun_out/pmc_kernel/pmc_mfma.json "$OUT/r3_pmc_mfma.json"
PMC_PROG='tools/fullcov_bench.py' PMC_KERNEL=fullcov_f16_kernel tools/pmc_kernel.sh > "$OUT/r3_pmc_fullcov.txt" 2>&1
PMC_PROG='tools/fullcov_bench.py --clustered' PMC_KERNEL=fullcov_f16_kernel tools/pmc_kernel.sh > "$OUT/r3_pmc_fullcov_clustered.txt" 2>&1
PMC_COUNTERS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" PMC_KERNEL=gmm1d \
  tools/pmc_generic.sh tools/gmm_bench.py --only gmm1d --chains 4 1024 --sorted > "$OUT/r3_pmc_gmm1d.txt" 2>&1
tools/chain_sweep.sh > "$OUT/r3_chain_sweep.txt" 2>&1
{ echo "# tools/gmm_bench.py, N = 100M, K = 8; rows in ascending order (what the host layer stores), then as drawn";
  timeout -k 10 200 python3 tools/gmm_bench.py --sorted --chains 1 2 3 4 16 64 1024;
  timeout -k 10 200 python3 tools/gmm_bench.py --chains 1 4 16 64 1024; } > "$OUT/r3_mixture_rows_ascending_output.txt" 2>&1
{ timeout -k 10 120 python3 tools/fullcov_bench.py; timeout -k 10 120 python3 tools/fullcov_bench.py --clustered; } > "$OUT/r3_fullcov_clustered_output.txt" 2>&1
ls -la "$OUT"
