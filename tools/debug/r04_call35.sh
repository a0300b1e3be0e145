cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
time bash tools/fuzz_unseen.sh 2979819 gpurun_out/r04/fuzz_unseen2.txt 1000 1100 1200 1300 1400 1500 1600 1700
