cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
(timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -k "merge or chain or x3 or e2e or border or graph or host_burst or sharded_hip or c5_geometry_48" -p no:cacheprovider) > gpurun_out/r04/t30.log 2>&1
tail -4 gpurun_out/r04/t30.log
for i in 1 2 3; do
  HHSR_MERGE_BORDER_SERIAL=1 python bench.py --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial', d['ms_per_step'], d['ms_per_step_eager'])"
  python bench.py --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forked', d['ms_per_step'], d['ms_per_step_eager'])"
done
