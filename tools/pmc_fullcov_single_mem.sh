#!/bin/bash
# L2 / memory counters of fullcov_single_kernel on rows stored cluster by cluster (two passes, each its own run).
set -e -o pipefail
export PMC_KERNEL=fullcov_single_kernel
PMC_COUNTERS="FETCH_SIZE GRBM_GUI_ACTIVE" tools/pmc_generic.sh tools/fullcov_bench.py --clustered
PMC_COUNTERS="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" tools/pmc_generic.sh tools/fullcov_bench.py --clustered
PMC_COUNTERS="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" tools/pmc_generic.sh tools/fullcov_bench.py --clustered
