#!/bin/bash
# SQ counter pass (clock, MFMA-busy share, waits) over the recurrent part of one encoder forward with the environment given on the command line:
#   tools/pmc_sq_gru.sh <tag> [VAR=value ...]      ->  gpurun_out/pmc_sq_<tag>.txt
export TMPDIR=/tmp
tag=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/pmc_$tag
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/gru_step_bench.py 8192 2 > gpurun_out/pmc_$tag.log 2>&1
python3 profiles/summarize.py sq gpurun_out/pmc_$tag/*/*counter_collection.csv gpurun_out/pmc_$tag/*/*kernel_trace.csv | grep -E "gru_|gemm_h3s|kernel  |#" > gpurun_out/pmc_sq_$tag.txt
rm -rf gpurun_out/pmc_$tag
cat gpurun_out/pmc_sq_$tag.txt
