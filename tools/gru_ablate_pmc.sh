#!/bin/bash
# SQ counter pass (clock, MFMA-busy share) over the recurrent part for ablation builds of the fused GRU step:
#   tools/gru_ablate_pmc.sh 0 256        (build/abl/lib_gruabl<N>.so, tools/build_abl.sh gemm_h3s TEPOSE_GRU_ABL gruabl 0 256)
export TMPDIR=/tmp
for v in "$@"; do
  export TEPOSE_AMD_LIB=$PWD/build/abl/lib_gruabl$v.so
  rm -rf gpurun_out/gruabl_$v; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/gruabl_$v -- python3 tools/gru_step_bench.py 8192 1 > gpurun_out/gruabl_$v.log 2>&1
  echo "== $v"; python3 profiles/summarize.py sq gpurun_out/gruabl_$v/*/*counter_collection.csv gpurun_out/gruabl_$v/*/*kernel_trace.csv
  rm -rf gpurun_out/gruabl_$v
done
