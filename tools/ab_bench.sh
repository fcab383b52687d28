#!/bin/bash
# Same-box A/B of library builds on the bench workload: tools/ab_bench.sh <lib.so|default> ...   (alternating, 2 rounds)
for round in 1 2; do
  for lib in "$@"; do
    if [ $lib = default ]; then unset TEPOSE_AMD_LIB; else export TEPOSE_AMD_LIB=$PWD/$lib; fi
    python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; g=d['roofline_gru_steps']
print('$lib  %.0f w/s  %.2f ms/step  gemm0 %.2f ms  gru %.2f ms' % (d['value'], d['ms_per_step'], r['avg_ms'], g['ms_per_forward']))"
  done
done
