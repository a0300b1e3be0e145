cd $GRAFT_REPO_ROOT
HHSR_LIB=$PWD/variants_pf5.so timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -k "grey or fft" -p no:cacheprovider 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
run() {
  L=$1; shift
  rm -rf /tmp/kt_$L
  env "$@" rocprofv3 --kernel-trace --stats -d /tmp/kt_$L -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-h2d --no-graph --steps 5 --warmup 2 --streams 1 > /tmp/kt_$L.log 2>&1
  echo "== $L"; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt_$L -name "*results.db" | head -1) 7 | grep "k_rows_fwd"
}
run default X=1
run pf5 HHSR_LIB=$GRAFT_REPO_ROOT/variants_pf5.so
run default2 X=1
run pf5b HHSR_LIB=$GRAFT_REPO_ROOT/variants_pf5.so
