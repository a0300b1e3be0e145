cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
bash tools/debug/kt_ab.sh "k_rows|k_cols|Total" fftbase fftslp fftnoc fftpow > gpurun_out/r04/kt_ab_fft_flags.txt 2>&1; cat gpurun_out/r04/kt_ab_fft_flags.txt
