export TMPDIR=/tmp
for gm in 2 4 8 16; do
  export TEPOSE_GRU_GM=$gm
  echo "== GM=$gm"; python3 tools/gru_step_bench.py 8192 6 2>&1 | grep "B="
done
for gm in 4 8 16; do
  export TEPOSE_GRU_GM=$gm
  d=gpurun_out/grutr_tmp; rm -rf $d
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $d -- python3 tools/gru_step_bench.py 8192 2 > $d.log 2>&1
  echo "== GM=$gm [FETCH_SIZE]"; python3 profiles/summarize.py pmcavg $d/*/*counter_collection.csv $d/*/*kernel_trace.csv | grep -A2 "gru_step16_kernel" | head -6
  rm -rf $d
done
