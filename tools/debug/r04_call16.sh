cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "robustness or rob_ or e2e_golden or batched" -p no:cacheprovider) > gpurun_out/r04/t16.log 2>&1
tail -4 gpurun_out/r04/t16.log
bash tools/debug/kt_ab.sh "k_rob_frames|Total" robbase rob8 rob6 rob4 > gpurun_out/r04/kt_ab_rob.txt 2>&1; cat gpurun_out/r04/kt_ab_rob.txt
