cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "x3 or c5_geometry or merge_golden or e2e_golden_scales" -p no:cacheprovider) > gpurun_out/r04/t7.log 2>&1
tail -4 gpurun_out/r04/t7.log
for v in xsbase xsnoedge xsedge; do echo "== $v"; HHSR_LIB=$PWD/variants_$v.so python tools/debug/border_cost.py 2>&1 | grep -v amdgpu; done > gpurun_out/r04/border_cost_ab.txt
cat gpurun_out/r04/border_cost_ab.txt
bash tools/debug/kt_ab.sh "k_align_wave<16, 1|Total" alignold alignnew > gpurun_out/r04/kt_ab_align.txt 2>&1
cat gpurun_out/r04/kt_ab_align.txt
python tools/debug/border_cost.py 20 3000 4000 2 > gpurun_out/r04/border_cost_x2.txt 2>&1; cat gpurun_out/r04/border_cost_x2.txt
