#!/bin/bash
# step time vs number of pipeline streams
for s in "$@"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --streams $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $s', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
