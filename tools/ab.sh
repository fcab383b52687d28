#!/bin/bash
# A/B of library builds: tools/ab.sh <lib1.so> <lib2.so> ...   (each: parity subset + bench line)
for lib in "$@"; do
  export TEPOSE_AMD_LIB=$PWD/$lib
  echo "== $lib"
  python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or golden or encoder or full_size" 2>&1 | tail -1
  for rep in 1 2; do
    python bench.py --batch 4096 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   value %.0f w/s  ms/step %.2f  gemm0 %.2f ms %.1f TF' % (d['value'], d['ms_per_step'], r['avg_ms'], r['achieved']))
"
  done
done
