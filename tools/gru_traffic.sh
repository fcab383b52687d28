#!/bin/bash
# Beyond-L2 traffic and L2 hit rate of the fused GRU step by build variant / XCD tile group (separate --pmc passes, kernel trace only):
#   tools/gru_traffic.sh <lib> [<lib> ...]      (each with TEPOSE_GRU_GM = 4 and 8)
export TMPDIR=/tmp
for lib in "$@"; do
  for gm in 4 8; do
    export TEPOSE_AMD_LIB=$lib TEPOSE_GRU_GM=$gm
    for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
      d=gpurun_out/grutr_tmp; rm -rf $d
      timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python3 tools/gru_step_bench.py 8192 2 > $d.log 2>&1
      echo "== $lib GM=$gm [$pass]"; python3 profiles/summarize.py pmcavg $d/*/*counter_collection.csv $d/*/*kernel_trace.csv | grep -A4 "gru_h3s16_kernel" | head -12
      rm -rf $d
    done
  done
done
