set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_gru_seq.py tests/test_gpu_geometry.py -x -q 2>&1 | tail -15
echo "== sweep default (persistent on)"
timeout 300 python tools/sweep.py 1x16 8x16 16x16 32x16 64x16 1x32 37x6 2>&1 | grep "B="
echo "== sweep persistent off"
TEPOSE_SEQ_MAX_M=0 timeout 300 python tools/sweep.py 1x16 8x16 16x16 32x16 64x16 1x32 37x6 2>&1 | grep "B="
echo "== sweep split_min_m=0 (B<=4 on the split path, persistent on)"
TEPOSE_SPLIT_MIN_M=0 timeout 300 python tools/sweep.py 1x16 1x32 4x16 2>&1 | grep "B="
echo "== sweep split_min_m=0 persistent off"
TEPOSE_SPLIT_MIN_M=0 TEPOSE_SEQ_MAX_M=0 timeout 300 python tools/sweep.py 1x16 1x32 4x16 2>&1 | grep "B="
