#!/bin/bash
# Same-box A/B of an environment knob on the bench workload: tools/ab_env.sh VAR v1 v2 ...   (alternating, 2 rounds)
var=$1; shift
for round in 1 2; do
  for v in "$@"; do
    env $var=$v python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; g=d['roofline_gru_steps']
print('$var=$v  %.0f w/s  %.2f ms/step  gemm0 %.2f ms  gru %.2f ms' % (d['value'], d['ms_per_step'], r['avg_ms'], g['ms_per_forward']))"
  done
done
