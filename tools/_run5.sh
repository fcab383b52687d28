cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gru_seq.py tests/test_gpu_geometry.py tests/test_gpu_parity.py tests/test_gpu_vibe.py tests/test_gpu_filters.py -x -q 2>&1 | tail -8
echo "== sweep"
python3 tools/sweep.py 1x16 1x32 4x16 8x16 16x16 32x16 64x16 37x6 2>&1 | grep "B="
echo "== reg_seq off"
TEPOSE_REG_SEQ_MAX_N=0 python3 tools/sweep.py 1x16 16x16 64x16 2>&1 | grep "B="
