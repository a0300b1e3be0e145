cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -k "merge or chain or e2e or host_burst or x3" -p no:cacheprovider) > gpurun_out/r04/t13.log 2>&1
tail -5 gpurun_out/r04/t13.log
bash tools/ab_variants.sh x2cls x2rgb x2cls x2rgb x2cls x2rgb > gpurun_out/r04/ab_x2_rgb.txt 2>&1; cat gpurun_out/r04/ab_x2_rgb.txt
