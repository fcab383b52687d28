#!/bin/bash
# Regenerate the rocprofv3 evidence for bench.py's workload (run on the GPU box through gpurun):
#   tools/profile.sh <tag>      -> gpurun_out/<tag>_{trace,fetch,write,sq}/...   then copy summaries to profiles/
# PMC passes are separate runs (never combined with the trace domains) and skip the small-batch extras,
# whose ~36k tiny launches are pathologically slow under per-dispatch counter readback.
set -e
tag=${1:-rNN}
export TMPDIR=/tmp
mkdir -p gpurun_out
common="--steps 2 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -- python3 bench.py $common > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_write -- python3 bench.py $common > gpurun_out/${tag}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${tag}_sq -- python3 bench.py $common > gpurun_out/${tag}_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/${tag}_lds -- python3 bench.py $common > gpurun_out/${tag}_lds.log 2>&1 || true
for d in trace fetch write sq lds; do ls gpurun_out/${tag}_$d/*/ | head -3; done
