#!/bin/bash
# SQ counter pass (clock, MFMA-busy share, wait share) of the persistent projection on the layer-0 shape for both MFMA shapes:
#   tools/shape_pmc.sh      (on the GPU box; TEPOSE_H3S=1: 32x32x16, 2: 16x16x32)
export TMPDIR=/tmp
for v in ${@:-1 2 1 2}; do
  export TEPOSE_H3S=$v
  rm -rf gpurun_out/shape_$v; timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/shape_$v -- python3 tools/h3_loop.py 131072 9216 2144 6 > gpurun_out/shape_$v.log 2>&1
  echo "== TEPOSE_H3S=$v"; python3 profiles/summarize.py sq gpurun_out/shape_$v/*/*counter_collection.csv gpurun_out/shape_$v/*/*kernel_trace.csv | grep -E "h3s|kernel"
  rm -rf gpurun_out/shape_$v
done
