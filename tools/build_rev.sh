#!/bin/bash
# Build the library from another git revision for same-box A/B runs:
#   tools/build_rev.sh <rev> <out.so> [extra hipcc flags]     (then TEPOSE_AMD_LIB=<out.so> python bench.py ...)
set -e
rev=$1; out=$(realpath -m $2); shift 2
tmp=$(mktemp -d)
mkdir -p $tmp/tepose_amd/csrc $tmp/include $(dirname $out)
for f in $(git ls-tree --name-only $rev tepose_amd/csrc/); do git show $rev:$f > $tmp/tepose_amd/csrc/$(basename $f); done
git show $rev:include/tepose_amd.h > $tmp/include/tepose_amd.h
(cd $tmp/tepose_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Xclang -target-feature -Xclang -packed-fp32-ops -DTEPOSE_NO_PACKED_FP32=1 "$@" -o $out *.hip)
rm -rf $tmp
echo built $out from $rev
