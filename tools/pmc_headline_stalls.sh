#!/bin/bash
# Two counter passes over the headline kernel (tools/kbench.py --f16 --reps 3): where its waves wait.
# Each pass is its own profiler run under its own time limit.
set -e -o pipefail
OUT=$PWD/gpurun_out/pmc_stalls
mkdir -p "$OUT"
PMC_KERNEL=linreg_f16_pipe PMC_COUNTERS="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES" \
  tools/pmc_generic.sh tools/kbench.py --f16 --reps 3 > "$OUT/pass1.txt" 2>&1
PMC_KERNEL=linreg_f16_pipe PMC_COUNTERS="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU" \
  tools/pmc_generic.sh tools/kbench.py --f16 --reps 3 > "$OUT/pass2.txt" 2>&1
PMC_KERNEL=linreg_f16_pipe PMC_COUNTERS="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
  tools/pmc_generic.sh tools/kbench.py --f16 --reps 3 > "$OUT/pass3.txt" 2>&1
cat "$OUT"/pass*.txt
