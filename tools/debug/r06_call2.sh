set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "grey" -p no:cacheprovider -s 2>&1 | tail -15 | tee gpurun_out/r06b/grey_tests.txt
timeout 300 python tools/fft_ab.py 2>&1 | tee gpurun_out/r06b/fft_ab.txt
timeout 300 python tools/fft_ab.py 6000 8000 2>&1 | tee -a gpurun_out/r06b/fft_ab.txt
timeout 300 python tools/fft_ab.py 3024 4032 0 3 2>&1 | tee -a gpurun_out/r06b/fft_ab.txt
