for S in 2 3 4; do for C in 3 4 5 7 8; do
  python bench.py --no-cpu-baseline --no-h2d --steps 15 --warmup 4 --streams $S --chunk $C 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $S chunk $C:', d['ms_per_step'], 'ms graph,', d['ms_per_step_eager'], 'ms eager')"
done; done
