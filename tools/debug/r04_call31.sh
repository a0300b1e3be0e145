cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
(timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "grey or fft or e2e_golden" -p no:cacheprovider) > gpurun_out/r04/t31.log 2>&1
tail -3 gpurun_out/r04/t31.log
cd /tmp && export TMPDIR=/tmp
for v in fft_old default; do
  if [ "$v" = default ]; then L=""; else L=$GRAFT_REPO_ROOT/variants_$v.so; fi
  rm -rf /tmp/kt_$v
  HHSR_LIB=$L rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-h2d --no-graph --steps 5 --warmup 2 --streams 1 > /tmp/kt_$v.log 2>&1
  echo "== $v"; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt_$v -name "*results.db" | head -1) 7 | grep "k_rows\|k_cols\|Total"
done
for i in 1 2; do for v in fft_old default; do
  if [ "$v" = default ]; then L=""; else L=$GRAFT_REPO_ROOT/variants_$v.so; fi
  HHSR_LIB=$L python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-h2d --steps 15 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v step', d['ms_per_step'])"
done; done
