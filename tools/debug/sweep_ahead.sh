# paced uploads with the copies of 0 / 1 / 2 chunks queued ahead, in the bench's process context (resident legs first)
export HHSR_PRELUDE=keep
for a in 0 1 2; do
  echo "== HHSR_UPLOAD_AHEAD=$a"
  HHSR_UPLOAD_AHEAD=$a HHSR_LEGS="pinned" python tools/debug/host_leg_timing.py 2>&1 | grep "pinned f32:\|pinned u16:"
done
echo "== bench, ahead 1"; HHSR_UPLOAD_AHEAD=1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['h2d']; print(d['ms_per_step'], h['ms_per_step_incl_h2d'], h['ms_per_step_incl_h2d_u16'], h['ms_per_step_numpy_pageable'])"
echo "== bench, ahead 0"; HHSR_UPLOAD_AHEAD=0 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['h2d']; print(d['ms_per_step'], h['ms_per_step_incl_h2d'], h['ms_per_step_incl_h2d_u16'], h['ms_per_step_numpy_pageable'])"
