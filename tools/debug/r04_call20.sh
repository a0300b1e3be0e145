cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
(timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -k "sharded or rccl or two_ranks or corners" -p no:cacheprovider) > gpurun_out/r04/t20.log 2>&1
tail -4 gpurun_out/r04/t20.log
