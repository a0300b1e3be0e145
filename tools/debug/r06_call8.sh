cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "denoiser or rob_sum or e2e_golden or merge_ref" -p no:cacheprovider 2>&1 | tail -6
python bench.py --no-cpu-baseline --no-c5 --steps 10 --warmup 4 --reps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ['ms_per_step','ms_per_step_denoiser','ms_per_step_weight_fp64','errors']})"
