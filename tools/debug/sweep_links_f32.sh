# chained-merge link placement on the host-resident float32 leg (chunks 4,4,4,3,2,1,1 -> chunk indices 0..6; default links 0-4)
export HHSR_LEGS="pinned f32:"
export HHSR_MERGE_CHAIN=1
for la in "0,1,2,3,4" "0,1,2,3,4,5" "1,3,4,5" "1,3,5" "2,4,5" "3,5" "1,2,3,4" "4,5" "2,4"; do
  echo "== links after chunks $la"
  HHSR_LINK_AFTER=$la python tools/debug/host_leg_timing.py 2>&1 | grep "pinned f32:"
done
