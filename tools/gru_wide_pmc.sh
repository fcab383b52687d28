#!/bin/bash
# (Needs the experiment wired in: header of tools/micro/gru_step16w_experiment.hip.)
# The wide-tile experiment's counter record: SQ pass (clock, MFMA-busy share, waits) and FETCH_SIZE / WRITE_SIZE passes over the recurrent
# part of one encoder forward, base step kernel and TEPOSE_GRU_WIDE=1, same box, separate --pmc passes with the kernel trace only.
#   tools/gru_wide_pmc.sh   ->  gpurun_out/r06_gru_wide_pmc.txt
export TMPDIR=/tmp
out=gpurun_out/r06_gru_wide_pmc.txt; : > $out
for wide in 0 1; do
  export TEPOSE_GRU_WIDE=$wide
  echo "== TEPOSE_GRU_WIDE=$wide  (timing, no profiler)" >> $out
  timeout 300 python3 tools/gru_step_bench.py 8192 6 2>/dev/null | grep "B=" >> $out
  d=gpurun_out/gw_tmp; rm -rf $d
  timeout 420 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $d -- python3 tools/gru_step_bench.py 8192 2 > $d.log 2>&1
  echo "-- SQ pass" >> $out
  python3 profiles/summarize.py sq $d/*/*counter_collection.csv $d/*/*kernel_trace.csv | grep -E "gru_step|kernel  |#" >> $out
  for pass in FETCH_SIZE WRITE_SIZE; do
    rm -rf $d
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python3 tools/gru_step_bench.py 8192 2 > $d.log 2>&1
    echo "-- $pass (per launch; FETCH_SIZE unit: 32 B on gfx950 = the x2 correction of MI355X_MICROARCH.md over the 64 B the tool assumes)" >> $out
    python3 profiles/summarize.py pmcavg $d/*/*counter_collection.csv $d/*/*kernel_trace.csv | grep -B1 -A3 "gru_step16" | head -16 >> $out
  done
  rm -rf $d $d.log
done
cat $out
