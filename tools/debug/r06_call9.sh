cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "robustness or rob_" -p no:cacheprovider 2>&1 | tail -3
bash tools/kernel_trace.sh r06e/kt 5 > /dev/null 2>&1
grep "k_rob_frames_row4\|hhsr kernels only" gpurun_out/r06e/kt/1stream.md | cut -c1-150
