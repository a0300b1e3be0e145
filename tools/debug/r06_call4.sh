cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
for i in 1 2; do
HHSR_FFT_STATIC=0 python bench.py --no-cpu-baseline --no-h2d --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('static=0', d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['sclk_mhz'], d['roofline']['avg_launch_ms'])"
python bench.py --no-cpu-baseline --no-h2d --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('static=3', d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['sclk_mhz'], d['roofline']['avg_launch_ms'])"
done
