cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -k "merge or x3 or c5_geometry or e2e or chain or c2_full or host_burst or sharded or two_ranks" -p no:cacheprovider) > gpurun_out/r04/t9.log 2>&1
tail -6 gpurun_out/r04/t9.log
for v in x2noedge x2edge; do echo "== $v"; HHSR_LIB=$PWD/variants_$v.so python tools/debug/border_cost.py 20 3000 4000 2 2>&1 | grep -v amdgpu | grep " ms"; done > gpurun_out/r04/border_cost_x2_ab.txt
cat gpurun_out/r04/border_cost_x2_ab.txt
bash tools/ab_variants.sh x2noedge x2edge x2noedge x2edge > gpurun_out/r04/ab_x2.txt 2>&1; cat gpurun_out/r04/ab_x2.txt
