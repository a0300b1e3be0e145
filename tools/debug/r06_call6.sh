cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "x3 or scales or merge_golden or xs or edge" -p no:cacheprovider 2>&1 | tail -4
for i in 1 2; do
python bench.py --no-cpu-baseline --no-h2d --steps 4 --warmup 3 --reps 3 --height 6000 --width 8000 --scale 3 --frames 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['sclk_mhz'], d['roofline']['avg_launch_ms'], d['roofline']['launch_ms_min_max'])"
done
