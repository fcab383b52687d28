cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== sweep"
python3 tools/sweep.py 1x16 1x32 4x16 8x16 16x16 32x16 64x16 37x6 2>&1 | grep "B="
echo "== sweep G0 skinny up to 1024"
TEPOSE_G0_SKINNY_MAX_M=1024 python3 tools/sweep.py 8x16 16x16 32x16 48x16 64x16 2>&1 | grep "B="
echo "== sweep G0 skinny off"
TEPOSE_G0_SKINNY_MAX_M=0 python3 tools/sweep.py 1x16 8x16 16x16 32x16 48x16 2>&1 | grep "B="
