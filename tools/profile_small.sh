#!/bin/bash
# rocprofv3 evidence for the small-batch shapes (BASELINE.json configs 1 / 2 / 5): kernel trace + FETCH_SIZE / WRITE_SIZE
# passes of tools/sweep.py with the shipped binary.   tools/profile_small.sh <tag>  ->  gpurun_out/<tag>_<shape>_{trace,fetch,write}
set -e
tag=${1:-rNN}
export TMPDIR=/tmp
mkdir -p gpurun_out
for sh in 1x16 64x16 1x32; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_${sh}_trace -- python3 tools/sweep.py $sh > gpurun_out/${tag}_${sh}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_${sh}_fetch -- python3 tools/sweep.py $sh > gpurun_out/${tag}_${sh}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_${sh}_write -- python3 tools/sweep.py $sh > gpurun_out/${tag}_${sh}_write.log 2>&1
done
