#!/bin/bash
# A/B of library variants (variants_<name>.so at the repo root, built with -D switches): step time + merge kernel time
for v in "$@"; do
  if [ "$v" = default ]; then L=""; else L=$PWD/variants_$v.so; fi
  HHSR_LIB=$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
