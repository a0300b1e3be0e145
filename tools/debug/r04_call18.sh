cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
HHSR_LIB=$PWD/variants_fftpf.so timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -k "grey or fft" -p no:cacheprovider 2>&1 | tail -2
bash tools/debug/kt_ab.sh "k_rows|Total" fftnopf fftpf > gpurun_out/r04/kt_ab_fft_pf.txt 2>&1
HHSR_FFT_PERSIST=512 bash tools/debug/kt_ab.sh "k_rows|Total" fftpf >> gpurun_out/r04/kt_ab_fft_pf.txt 2>&1
HHSR_FFT_PERSIST=1024 bash tools/debug/kt_ab.sh "k_rows|Total" fftpf >> gpurun_out/r04/kt_ab_fft_pf.txt 2>&1
grep -v "== default" gpurun_out/r04/kt_ab_fft_pf.txt | cut -c1-150
