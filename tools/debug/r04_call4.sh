cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -k "x3 or c5_geometry_48 or merge_golden or e2e_golden_scales" -p no:cacheprovider) > gpurun_out/r04/t4.log 2>&1
tail -3 gpurun_out/r04/t4.log
bash tools/debug/ab_c5.sh x3old x3nodb x3db > gpurun_out/r04/ab_c5_b.txt 2>&1
cat gpurun_out/r04/ab_c5_b.txt
./tools/ubench/valu_occupancy > gpurun_out/r04/valu_occupancy.txt 2>&1
cat gpurun_out/r04/valu_occupancy.txt
