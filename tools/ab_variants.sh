#!/bin/bash
# A/B of library variants: merge kernel time + step time
for v in base db occ5 dbocc5 occ6 occ3; do
  HHSR_LIB=$PWD/variants_$v.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
