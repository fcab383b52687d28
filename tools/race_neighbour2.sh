#!/bin/bash
# tools/race_neighbour2.sh <lib.so> [modes...]: the stand-alone packed skinning kernel (build/micro/skin4_v0, checked) next to
# tools/race_neighbour.py <mode> in a second process
export TEPOSE_AMD_LIB=$1; shift
for m in ${@:-gemm memset gemm_h3s gemm_f32 matmul matmul16 fill}; do
  python tools/race_neighbour.py $PWD 9 $m > /tmp/nb.log 2>&1 &
  sleep 5.5
  build/micro/skin4_v0 6000 20 2>&1 | tail -n 1 | sed "s/^/neighbour $m: /"
  wait
  tail -n 1 /tmp/nb.log
done
