#!/bin/bash
# tools/race_neighbour.sh <lib.so> [modes...]: race_probe_smpl.py (checked) next to race_neighbour.py <mode> (second process)
export TEPOSE_AMD_LIB=$1; shift
for m in ${@:-smpl fill equal gemm aa d2h idle}; do
  python tools/race_neighbour.py $PWD 14 $m > /tmp/nb.log 2>&1 &
  sleep 5
  python tools/race_probe_smpl.py $PWD 4000 2>&1 | tail -n 1 | sed "s/^/neighbour $m: /"
  wait
  tail -n 1 /tmp/nb.log
done
