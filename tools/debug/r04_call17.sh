cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
bash tools/debug/kt_ab.sh "k_rows|k_cols|Total" fftw5 fftw4 > gpurun_out/r04/kt_ab_fft_wpe.txt 2>&1
HHSR_FFT_PERSIST=512 bash tools/debug/kt_ab.sh "k_rows|k_cols|Total" fftw4 >> gpurun_out/r04/kt_ab_fft_wpe.txt 2>&1
cat gpurun_out/r04/kt_ab_fft_wpe.txt
