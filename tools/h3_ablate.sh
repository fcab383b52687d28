#!/bin/bash
# Timing-only ablation builds of the split GEMM (build/abl/lib_h3abl<N>.so = -DTEPOSE_H3_ABL=N) on the layer-0
# projection shape (N is a bit mask: 1 no DMA, 2 no barrier, 4 no fragment reads), with the SQ counter pass that gives clock and MFMA-busy share.  Run on the GPU box.
export TMPDIR=/tmp
for v in ${@:-base 1 4 5}; do
  if [ $v = base ]; then unset TEPOSE_AMD_LIB; else export TEPOSE_AMD_LIB=$PWD/build/abl/lib_h3abl$v.so; fi
  rm -rf gpurun_out/abl_$v; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/abl_$v -- python3 tools/h3_loop.py 131072 9216 2144 4 > gpurun_out/abl_$v.log 2>&1
  echo "== $v"; python3 profiles/summarize.py sq gpurun_out/abl_$v/*/*counter_collection.csv gpurun_out/abl_$v/*/*kernel_trace.csv | grep h3
done
