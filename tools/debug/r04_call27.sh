cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
(timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "edge_frames_all_sides" -p no:cacheprovider) 2>&1 | tail -25
