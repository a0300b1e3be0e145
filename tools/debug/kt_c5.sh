cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_c5
rocprofv3 --kernel-trace --stats -d /tmp/kt_c5 -o kt -- python $GRAFT_REPO_ROOT/bench.py --reps 1 --no-cpu-baseline --no-h2d --no-graph --steps 3 --warmup 1 --streams 1 --height 6000 --width 8000 --scale 3 > /tmp/kt_c5.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt_c5 -name "*results.db" | head -1) 4 | grep -v "at::native\|Cijk\|rocclr" | head -24
