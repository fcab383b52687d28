#!/bin/bash
# Timing-only ablations of the fused GRU step (gru_h3s16_kernel<0,2>): what each part of the kernel costs.  WRONG results by construction.
#   tools/build_abl.sh gemm_h3s16 TEPOSE_G16_ABL g16abl 0 1 2 4 8 12 16 17 64 65 81     (here, on the build box)
#   tools/g16_ablate.sh                                                                 (on the GPU box)
for r in 1 2; do
  for v in 0 1 2 4 8 12 16 17 64 65 81; do
    echo "== ABL $v: $(TEPOSE_AMD_LIB=build/abl/lib_g16abl$v.so python3 tools/gru_step_bench.py 8192 3 2>&1 | tail -n 1)"
  done
done
