cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt64
rocprofv3 --kernel-trace --stats -d /tmp/kt64 -o kt -- python $GRAFT_REPO_ROOT/bench.py --reps 1 --no-cpu-baseline --no-h2d --no-graph --weight-fp64 --steps 3 --warmup 1 --streams 1 > /tmp/kt64.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r06g
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt64 -name "*results.db" | head -1) 4 | grep -v "at::native\|Cijk\|rocclr" | head -24 > $GRAFT_REPO_ROOT/gpurun_out/r06g/kernel_trace_weight_fp64.md
cat $GRAFT_REPO_ROOT/gpurun_out/r06g/kernel_trace_weight_fp64.md | cut -c1-150
