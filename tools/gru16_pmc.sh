#!/bin/bash
# SQ counter pass (clock, MFMA-busy share, waits) of the recurrent part for both MFMA shapes of the fused GRU step:
#   tools/gru16_pmc.sh        (TEPOSE_MFMA16=0: 32x32x16, 2: 16x16x32)
export TMPDIR=/tmp
for v in 0 5; do
  export TEPOSE_MFMA16=$v
  rm -rf gpurun_out/gru16_$v; timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/gru16_$v -- python3 tools/gru_step_bench.py 8192 2 > gpurun_out/gru16_$v.log 2>&1
  echo "== TEPOSE_MFMA16=$v"; python3 profiles/summarize.py sq gpurun_out/gru16_$v/*/*counter_collection.csv gpurun_out/gru16_$v/*/*kernel_trace.csv | grep -E "gru_h3s|gemm_h3s_kernel|kernel  "
  rm -rf gpurun_out/gru16_$v
done
