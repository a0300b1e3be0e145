cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
(timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "flow_bound_retry" -p no:cacheprovider) 2>&1 | tail -30
