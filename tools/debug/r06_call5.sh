cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "merge or x2 or chain or e2e or graph" -p no:cacheprovider 2>&1 | tail -5
for i in 1 2; do
python bench.py --no-cpu-baseline --no-h2d --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['sclk_mhz'], d['roofline']['avg_launch_ms'], d['roofline']['launch_ms_min_max'])"
done
