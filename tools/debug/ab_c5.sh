#!/bin/bash
# C5 geometry (48 MP x 20 x3) with library variants: step time + k_merge_xs<3> launch time
for v in "$@"; do
  if [ "$v" = default ]; then L=""; else L=$PWD/variants_$v.so; fi
  HHSR_LIB=$L python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-h2d --height 6000 --width 8000 --scale 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"
done
