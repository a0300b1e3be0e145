cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "grey or fft or e2e_golden or pyramid" -p no:cacheprovider) > gpurun_out/r04/t10.log 2>&1
tail -4 gpurun_out/r04/t10.log
bash tools/debug/kt_ab.sh "k_rows|k_cols|Total" fftnopow fftpow > gpurun_out/r04/kt_ab_fft_c3.txt 2>&1; cat gpurun_out/r04/kt_ab_fft_c3.txt
cd /tmp && export TMPDIR=/tmp
for v in fftnopow fftpow; do
  rm -rf /tmp/kt_c5_$v
  HHSR_LIB=$GRAFT_REPO_ROOT/variants_$v.so rocprofv3 --kernel-trace --stats -d /tmp/kt_c5_$v -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-h2d --no-graph --steps 3 --warmup 1 --streams 1 --height 6000 --width 8000 --scale 3 > /tmp/kt_c5_$v.log 2>&1
  echo "== $v"; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt_c5_$v -name "*results.db" | head -1) 4 | grep -E "k_rows|k_cols|Total|k_merge_xs" | cut -c1-150
done > $GRAFT_REPO_ROOT/gpurun_out/r04/kt_ab_fft_c5.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r04/kt_ab_fft_c5.txt
