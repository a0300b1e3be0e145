cd /tmp && export TMPDIR=/tmp
for v in default map1 map2; do
  if [ "$v" = default ]; then L=""; else L=$GRAFT_REPO_ROOT/variants_$v.so; fi
  rm -rf /tmp/kt_$v
  HHSR_LIB=$L rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-h2d --no-graph --steps 5 --warmup 2 --streams 1 > /tmp/kt_$v.log 2>&1
  echo "== $v"; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt_$v -name "*results.db" | head -1) 7 | grep "k_rows\|k_cols\|Total"
done
