#!/bin/bash
# The round's measurements without the test suite and the full-size oracle runs (tools/collect_round.sh has those).
# usage: tools/collect_round_lite.sh <name> [what...]   what: pmc_x2 pmc_x3 bench traces pmc_all emulate (default: all)
NAME=${1:-r06}; shift
WHAT=${*:-"pmc_x2 pmc_x3 bench traces pmc_all emulate"}
OUT=gpurun_out/$NAME
mkdir -p $OUT
CS=handheld-multi-frame-super-resolution_amd/csrc
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has pmc_x2; then
  bash tools/pmc_merge.sh "k_merge_x2" $NAME/pmc_x2 --reps 1 --no-h2d --steps 1 --warmup 0 > /dev/null 2>&1
  python tools/pmc_report.py $OUT/pmc_x2 k_merge_x2 profiles/${NAME}_pmc_merge.json $CS/hhsr_merge.h,$CS/hhsr_merge_x2.hip "3000x4000x20 x2" \
    "tools/pmc_merge.sh k_merge_x2 (bench.py --no-cpu-baseline --no-h2d --steps 1 --warmup 0), profiles/${NAME}_pmc_merge_x2.md" > $OUT/pmc_x2.md
  cp profiles/${NAME}_pmc_merge.json $OUT/pmc_merge.json
fi
if has pmc_x3; then
  bash tools/pmc_merge.sh "k_merge_xs" $NAME/pmc_x3 --reps 1 --no-h2d --steps 1 --warmup 0 --height 6000 --width 8000 --scale 3 > /dev/null 2>&1
  python tools/pmc_report.py $OUT/pmc_x3 k_merge_xs profiles/${NAME}_pmc_merge_x3.json $CS/hhsr_merge.h,$CS/hhsr_merge_xs.hip "6000x8000x20 x3" \
    "tools/pmc_merge.sh k_merge_xs (bench.py --no-cpu-baseline --no-h2d --steps 1 --warmup 0 --height 6000 --width 8000 --scale 3), profiles/${NAME}_pmc_merge_x3.md" > $OUT/pmc_x3.md
  cp profiles/${NAME}_pmc_merge_x3.json $OUT/pmc_merge_x3.json
fi
if has bench; then
  T0=$(date +%s); python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench.py default run: $(( $(date +%s) - T0 )) s wall" > $OUT/bench_n1.time
  python bench.py --height 6000 --width 8000 --scale 3 --frames 20 --steps 5 --warmup 2 --no-cpu-baseline --no-h2d > $OUT/bench_c5.json 2> /dev/null
  python bench.py --frames 8 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d > $OUT/bench_c2.json 2> /dev/null
fi
if has traces; then
  bash tools/kernel_trace.sh $NAME/kt 5 > /dev/null 2>&1
  bash tools/debug/kt_c5.sh 2>&1 | grep -v amdgpu.ids > $OUT/kernel_trace_c5.md
fi
if has pmc_all; then bash tools/pmc_all.sh $NAME/pmc_all > /dev/null 2>&1; fi
if has emulate; then
  python tools/debug/emulate_ranks.py --worlds 1,2,4,8 --steps 10 2>&1 | grep "^{" > $OUT/emulate_ranks_c3.jsonl
  python tools/debug/emulate_ranks.py --worlds 1,2,4,8 --steps 3 --height 6000 --width 8000 --scale 3 --strategies rows 2>&1 | grep "^{" > $OUT/emulate_ranks_c5.jsonl
fi
find $OUT -name "*agent_info*" -delete
cut -c1-400 $OUT/bench_n1.json 2>/dev/null
