cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "align or bm_ or ica or e2e_golden or upscale" -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_blocks'], d['spread_pct'], d['sclk_mhz_blocks'])"; done
bash tools/kernel_trace.sh r06f/kt 5 > /dev/null 2>&1
grep "k_align_wave\|hhsr kernels only" gpurun_out/r06f/kt/1stream.md | cut -c1-150
