#!/bin/bash
# Timing-only ablation builds of the fused GRU step (build/abl/lib_gruabl<N>.so = gemm_h3s.hip with -DTEPOSE_GRU_ABL=N; bit mask:
# 1 no LDS-DMA, 2 no epilogue loads, 4 no epilogue stores, 8 no LDS turn, 16 no MFMA, 64 no epilogue).  Run on the GPU box:
#   tools/gru_ablate.sh [base 1 2 ...]
for v in ${@:-base 64 2 4 6 8 1 16 65 17 81 base}; do
  if [ $v = base ]; then unset TEPOSE_AMD_LIB; else export TEPOSE_AMD_LIB=$PWD/build/abl/lib_gruabl$v.so; fi
  echo "== $v: $(python3 tools/gru_step_bench.py 8192 6 2>&1 | tail -1)"
done
