cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for L in "" $PWD/variants_nw2.so; do
  echo "== lib '$L'"
  HHSR_LIB=$L timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "x3 or scales or c5_geometry or merge_golden" -p no:cacheprovider 2>&1 | tail -3
done
bash tools/debug/ab_c5.sh default nw2 default nw2
