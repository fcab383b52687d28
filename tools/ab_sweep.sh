#!/bin/bash
# Same-box A/B of library builds on tools/sweep.py shapes: tools/ab_sweep.sh "<shapes>" <lib.so|default> ...
shapes=$1; shift
for lib in "$@"; do
  if [ $lib = default ]; then unset TEPOSE_AMD_LIB; else export TEPOSE_AMD_LIB=$PWD/$lib; fi
  echo "== $lib"; python tools/sweep.py $shapes 2>&1 | grep "B="
done
