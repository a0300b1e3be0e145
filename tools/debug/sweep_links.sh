# chained-merge link placement on the host-resident uint16 leg (chunks 4,4,4,3,2,1,1 -> chunk indices 0..6)
export HHSR_LEGS="pinned u16"
echo "== no chain"; python tools/debug/host_leg_timing.py 2>&1 | grep "pinned u16:"
for la in "0,1,2,3,4" "2" "1" "1,3" "2,4" "3" "0,2,4" "1,2,3"; do
  echo "== chain, links after chunks $la"
  HHSR_MERGE_CHAIN=1 HHSR_LINK_AFTER=$la python tools/debug/host_leg_timing.py 2>&1 | grep "pinned u16:"
done
