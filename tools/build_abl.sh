#!/bin/bash
# Build timing-only ablation variants of one source file against the cached objects of the others:
#   tools/build_abl.sh <file-stem> <macro> <name-prefix> v1 v2 ...   ->  build/abl/lib_<prefix><v>.so  (-D<macro>=<v>)
# (objects of the unmodified sources: build/obj/*.o, rebuilt here when missing)
set -e
stem=$1; macro=$2; prefix=$3; shift 3
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DTEPOSE_NO_PACKED_FP32=1 -Wno-inline-asm"
mkdir -p build/obj build/abl
for f in gemm gemm_h3 gemm_h3s gemm_h3s16c gru_step16 skinny skinny_h3 gru_seq reg_seq misc smpl metrics filters api; do
  if [ ! -f build/obj/$f.o ] || [ tepose_amd/csrc/$f.hip -nt build/obj/$f.o ] || [ tepose_amd/csrc/common.h -nt build/obj/$f.o ]; then
    (cd tepose_amd/csrc && hipcc $FLAGS -c $f.hip -o ../../build/obj/$f.o 2>&1 | grep -v packed-fp32 || true) &
  fi
done
wait
for v in "$@"; do
  (cd tepose_amd/csrc && hipcc $FLAGS -D$macro=$v -c $stem.hip -o ../../build/abl/${stem}_$v.o 2>&1 | grep -v packed-fp32 || true
   cd ../.. && objs=""; for f in gemm gemm_h3 gemm_h3s gemm_h3s16c gru_step16 skinny skinny_h3 gru_seq reg_seq misc smpl metrics filters api; do
     if [ $f = $stem ]; then objs="$objs build/abl/${stem}_$v.o"; else objs="$objs build/obj/$f.o"; fi; done
   hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/lib_$prefix$v.so $objs) &
done
wait
ls -la build/abl/lib_$prefix*.so
