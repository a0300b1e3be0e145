cd /tmp && export TMPDIR=/tmp
run() {  # label, env...
  L=$1; shift
  rm -rf /tmp/kt_$L
  env "$@" rocprofv3 --kernel-trace --stats -d /tmp/kt_$L -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-h2d --no-graph --steps 5 --warmup 2 --streams 1 > /tmp/kt_$L.log 2>&1
  echo "== $L"; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt_$L -name "*results.db" | head -1) 7 | grep "k_rows\|k_cols\|Total"
}
run default X=1
run nc1 HHSR_FFT_NC=1
