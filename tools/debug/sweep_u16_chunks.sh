# host-resident uint16 leg: early small chunks (the GPU starts sooner) with / without one chained-merge link
export HHSR_LEGS="pinned u16:"
run() { echo "== chunks $1 chain=$2 links=$3"; HHSR_HOST_CHUNKS=$1 HHSR_MERGE_CHAIN=$2 HHSR_LINK_AFTER=$3 python tools/debug/host_leg_timing.py 2>&1 | grep "pinned u16:"; }
run "4,4,4,3,2,1,1" 0 ""
run "1,2,4,4,4,2,1,1" 0 ""
run "2,4,4,4,3,1,1" 0 ""
run "1,2,4,4,4,2,1,1" 1 "3"
run "1,2,4,4,4,2,1,1" 1 "2"
run "1,2,4,4,4,2,1,1" 1 "2,4"
run "2,4,4,4,3,1,1" 1 "2"
run "2,3,3,3,3,3,1,1" 1 "3"
run "1,1,2,3,4,4,2,1,1" 1 "4"
