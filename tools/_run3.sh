cd $GRAFT_REPO_ROOT
export TEPOSE_SPLIT_MIN_M=0
for m in 0 1 2 3 4 12 13 15 16; do
  echo "== abl $m"
  TEPOSE_AMD_LIB=$PWD/build/abl/seq_$m.so python3 tools/sweep.py 1x16 16x16 64x16 2>&1 | grep "B="
done
