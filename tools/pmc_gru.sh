#!/bin/bash
# Counter passes over the recurrent part (tools/gru_step_bench.py) for the kernels that contain "gru" / "gemm_h3s":
#   tools/pmc_gru.sh <tag> [env assignments for the run, e.g. TEPOSE_GRU_STATE=fp32]
# Separate --pmc passes (never with the trace domains); digest by profiles/summarize.py pmcavg.  Every pass runs under its own
# `timeout`: a TA_* counter set (TA_BUSY_avr, TA_*_STALLED_BY_TC_CYCLES_sum) did not finish in 20 minutes on this pool and was
# dropped from the list -- do not put derived TA metrics back without a bound.
tag=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL GRBM_GUI_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/${tag}_p$i
  timeout 420 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${tag}_p$i -- python3 tools/gru_step_bench.py 8192 2 > gpurun_out/${tag}_p$i.log 2>&1
  python3 profiles/summarize.py pmcavg gpurun_out/${tag}_p$i/*/*counter_collection.csv gpurun_out/${tag}_p$i/*/*kernel_trace.csv
done
