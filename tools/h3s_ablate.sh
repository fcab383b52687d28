#!/bin/bash
# Timing-only ablation builds of the persistent projection (build/abl/lib_h3sabl<N>.so = gemm_h3s.hip with -DTEPOSE_H3S_ABL=N;
# bit mask: 1 no LDS-DMA after a tile's first three stages, 4 fragment reads only in a tile's first K-tile, 8 no C stores) on the
# layer-0 projection shape, with the SQ counter pass that gives clock and MFMA-busy share.  The ring keeps REAL operand data in
# every variant (the first stages of each tile are always fetched), so the matrix pipes' power -- these kernels are power-limited
# -- stays comparable; compare clk_GHz between rows before reading a time difference as "the cost of X".
#   tools/build_abl.sh gemm_h3s TEPOSE_H3S_ABL h3sabl 0 1 4 5 8 13 ; tools/h3s_ablate.sh   (on the GPU box)
export TMPDIR=/tmp TEPOSE_H3S=1
for v in ${@:-0 1 4 5 8 13}; do
  export TEPOSE_AMD_LIB=$PWD/build/abl/lib_h3sabl$v.so
  rm -rf gpurun_out/h3sabl_$v; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/h3sabl_$v -- python3 tools/h3_loop.py 131072 9216 2144 6 > gpurun_out/h3sabl_$v.log 2>&1
  echo "== $v"; python3 profiles/summarize.py sq gpurun_out/h3sabl_$v/*/*counter_collection.csv gpurun_out/h3sabl_$v/*/*kernel_trace.csv | grep -E "h3s|kernel"
  rm -rf gpurun_out/h3sabl_$v
done
