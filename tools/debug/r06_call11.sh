cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_blocks'], d['spread_pct'], d['sclk_mhz'], d['sclk_measured_in'][-60:], d['timed_region_s'])"; done
