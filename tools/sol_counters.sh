#!/bin/bash
# Speed-of-light counters of single convolution layers: one rocprofv3 --pmc pass per counter group over
# tools/conv_bench.py restricted to one layer and one tile configuration, kept per dispatch, then tools/sol_table.py turns the
# passes into a per-chunk table (matrix / vector / LDS / vector-memory cycles per CU against the kernel's wall time).
#   bash tools/sol_counters.sh <tag> <cfg> "<layer substring>"      -> gpurun_out/sol_<tag>_<group>_dispatches.csv
set -o pipefail
tag=$1; cfg=$2; match=$3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
run() {   # group name, counters
    bash tools/prof_pmc_script.sh sol_${tag}_$1 "$2" tools/conv_bench.py --cfgs $cfg --match "$match" --iters 6 || echo "pass $1 failed"
}
run sq1 "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA"
run sq2 "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES"
run sq3 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32"
# (the TA_* / TCP_* / TD_* counters are not collected: a pass with TA_TA_BUSY_sum aborts inside rocprofv3 on this image -- signal 6 from
#  its counter setup, round 4's first measurement call -- and then hangs until the box's silence limit; vector-memory pressure is read from
#  SQ_ACTIVE_INST_VMEM / SQ_INST_CYCLES_VMEM_RD instead)
run grbm "GRBM_GUI_ACTIVE FETCH_SIZE"
ls gpurun_out | grep "sol_${tag}_" | head -20
