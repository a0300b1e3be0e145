#!/bin/bash
# Everything the round's profiles / PARITY.md are made from, in one GPU call.  usage: tools/collect_round.sh <name>
NAME=${1:-r05}
OUT=gpurun_out/$NAME
mkdir -p $OUT
HHSR_PARITY_LOG=$PWD/$OUT/parity.jsonl timeout 2400 python -m pytest tests -q -m gpu --timeout 1200 > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
python tools/parity_report.py $OUT/parity.jsonl > $OUT/PARITY.md
# the merge kernel's counters first: bench.py's roofline.traffic reads the record (keyed to the kernel source's hash)
bash tools/pmc_merge.sh "k_merge_x2" $NAME/pmc_x2 --no-h2d --steps 1 --warmup 0 > /dev/null 2>&1
CS=handheld-multi-frame-super-resolution_amd/csrc
MSRC=$CS/hhsr_merge.h,$CS/hhsr_merge_x2.hip
python tools/pmc_report.py $OUT/pmc_x2 k_merge_x2 profiles/${NAME}_pmc_merge.json $MSRC "3000x4000x20 x2" \
  "tools/pmc_merge.sh k_merge_x2 (bench.py --no-cpu-baseline --no-h2d --steps 1 --warmup 0), profiles/${NAME}_pmc_merge_x2.md" > $OUT/pmc_x2.md
cp profiles/${NAME}_pmc_merge.json $OUT/pmc_merge.json
# ... and the x3 kernel's on the C5 geometry (48 MP x 20 x3 on one GPU)
bash tools/pmc_merge.sh "k_merge_xs" $NAME/pmc_x3 --no-h2d --steps 1 --warmup 0 --height 6000 --width 8000 --scale 3 > /dev/null 2>&1
python tools/pmc_report.py $OUT/pmc_x3 k_merge_xs profiles/${NAME}_pmc_merge_x3.json $CS/hhsr_merge.h,$CS/hhsr_merge_xs.hip "6000x8000x20 x3" \
  "tools/pmc_merge.sh k_merge_xs (bench.py --no-cpu-baseline --no-h2d --steps 1 --warmup 0 --height 6000 --width 8000 --scale 3), profiles/${NAME}_pmc_merge_x3.md" > $OUT/pmc_x3.md
cp profiles/${NAME}_pmc_merge_x3.json $OUT/pmc_merge_x3.json
T0=$(date +%s); python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench.py default run: $(( $(date +%s) - T0 )) s wall" > $OUT/bench_n1.time
python bench.py --height 6000 --width 8000 --scale 3 --frames 20 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c5.json 2> /dev/null
python bench.py --frames 8 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c2.json 2> /dev/null
bash tools/kernel_trace.sh $NAME/kt 5 > /dev/null 2>&1
bash tools/pmc_all.sh $NAME/pmc_all > /dev/null 2>&1
# host-resident legs, the copy / kernel timelines of one host-resident step, measured bounds
python tools/debug/host_leg_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/host_legs.txt
bash tools/debug/h2d_trace.sh f32 2>&1 | grep -v "^\[" | grep -v "^\['id'" > $OUT/h2d_trace_f32.txt
bash tools/debug/h2d_trace.sh u16 2>&1 | grep -v "^\[" | grep -v "^\['id'" > $OUT/h2d_trace_u16.txt
python tools/debug/mono_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/mono_timing.txt
bash tools/debug/kt_mono.sh 2>&1 | grep -v amdgpu.ids > $OUT/kernel_trace_mono.md
bash tools/debug/kt_c5.sh 2>&1 | grep -v amdgpu.ids > $OUT/kernel_trace_c5.md
# per-rank compute of the multi-GPU strategies on this one GPU (tools/debug/emulate_ranks.py)
python tools/debug/emulate_ranks.py --worlds 1,2,4,8 --steps 10 2>&1 | grep "^{" > $OUT/emulate_ranks_c3.jsonl
python tools/debug/emulate_ranks.py --worlds 2,4,8 --steps 10 --strategies rows --stage-frames 4 2>&1 | grep "^{" > $OUT/emulate_ranks_c3_staged.jsonl
python tools/debug/emulate_ranks.py --worlds 1,2,4,8 --steps 3 --height 6000 --width 8000 --scale 3 --strategies rows 2>&1 | grep "^{" > $OUT/emulate_ranks_c5.jsonl
# round 5: the overlap changes of the per-rank step A/B; the headline burst and the C5 geometry against the oracle at full size
# (two-sided; minutes of all host cores).  (The micro-benchmarks and probes of round 4 — VALU rates, occupancy, LDS patterns,
# counter calibration, hardware queues, border cost, covariance-inline bound — did not change: profiles/r04_*.)
bash tools/debug/ab_rows_overlap.sh > $OUT/rows_overlap_ab.txt 2>&1
python tools/full_size_oracle.py --workers 8 --out $OUT/c3_full_oracle.txt > /dev/null 2>&1
python tools/full_size_oracle.py --height 3000 --width 8000 --frames 5 --scale 3 --workers 4 --out $OUT/c5_geometry_oracle.txt > /dev/null 2>&1
find $OUT -name "*agent_info*" -delete
cut -c1-600 $OUT/bench_n1.json
