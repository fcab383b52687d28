cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sh in 1x16 16x16 64x16; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02b_$sh -- python3 tools/sweep.py $sh > gpurun_out/r02b_$sh.log 2>&1
  f=$(ls gpurun_out/r02b_$sh/*/*kernel_trace.csv | head -1)
  echo "== $sh"; grep "B=" gpurun_out/r02b_$sh.log
  python3 profiles/summarize.py trace $f | head -24
done
