cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "grey or fft or e2e_golden or c5_geometry or c2_full" -p no:cacheprovider 2>&1 | tail -3
for i in 1 2 3; do
  HHSR_FFT_NT_ROWS=512 python bench.py --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows512', d['ms_per_step'], d['ms_per_step_eager'])"
  python bench.py --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows256', d['ms_per_step'], d['ms_per_step_eager'])"
done
bash tools/debug/ab_c5.sh default
