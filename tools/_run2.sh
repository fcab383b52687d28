cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export TEPOSE_SPLIT_MIN_M=0
for sh in 1x16 64x16; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02a_$sh -- python3 tools/sweep.py $sh > gpurun_out/r02a_$sh.log 2>&1
  f=$(ls gpurun_out/r02a_$sh/*/*kernel_trace.csv | head -1)
  echo "== $sh"; grep "B=" gpurun_out/r02a_$sh.log
  python3 profiles/summarize.py trace $f | head -40
done
