cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "x3 or c5_geometry or merge_golden or e2e_golden_scales" -p no:cacheprovider) > gpurun_out/r04/t8.log 2>&1
tail -4 gpurun_out/r04/t8.log
python tools/debug/border_cost.py 2>&1 | grep -v amdgpu > gpurun_out/r04/border_cost_c.txt; cat gpurun_out/r04/border_cost_c.txt
bash tools/debug/ab_c5.sh default > gpurun_out/r04/ab_c5_e.txt 2>&1; cat gpurun_out/r04/ab_c5_e.txt
