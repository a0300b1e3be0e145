cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(time timeout 2400 python -m pytest tests/test_hip_parity.py -q -m gpu -k "x3 or c5_geometry or merge_golden or e2e_golden_scales or e2e_scales or mono" -p no:cacheprovider) > gpurun_out/r04/t3.log 2>&1
tail -8 gpurun_out/r04/t3.log
bash tools/debug/ab_c5.sh x3old x3new > gpurun_out/r04/ab_c5_a.txt 2>&1
cat gpurun_out/r04/ab_c5_a.txt
