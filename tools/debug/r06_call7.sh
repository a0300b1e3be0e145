cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
(time timeout 1700 python -m pytest tests/test_fuzz_findings.py tests/test_fuzz_parity.py tests/test_abi.py "tests/test_hip_parity.py::test_c3_full_size_two_sided_four_frames" -x -q -m gpu -p no:cacheprovider -s --durations=8 2>&1 | tail -30) 2>&1 | tee gpurun_out/r06d/new_tests.txt
