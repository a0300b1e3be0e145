cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
HHSR_FFT_RB=1 HHSR_FFT_NC=1 HHSR_FFT_PERSIST=1280 HHSR_LIB=$PWD/variants_nt256.so timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -k "grey or fft" -p no:cacheprovider 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
run() {  # label lib persist
  rm -rf /tmp/kt_$1
  HHSR_FFT_PERSIST=$3 HHSR_LIB=$2 rocprofv3 --kernel-trace --stats -d /tmp/kt_$1 -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-h2d --no-graph --steps 5 --warmup 2 --streams 1 > /tmp/kt_$1.log 2>&1
  echo "== $1"; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt_$1 -name "*results.db" | head -1) 7 | grep "k_rows\|k_cols\|Total"
}
run default "" 768
export HHSR_FFT_RB=1 HHSR_FFT_NC=1
run nt256_1280 $GRAFT_REPO_ROOT/variants_nt256.so 1280
run nt256_1024 $GRAFT_REPO_ROOT/variants_nt256.so 1024
run nt256_1536 $GRAFT_REPO_ROOT/variants_nt256.so 1536
