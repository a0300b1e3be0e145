cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_mono
rocprofv3 --kernel-trace --stats -d /tmp/kt_mono -o kt -- python $GRAFT_REPO_ROOT/tools/debug/mono_timing.py > /tmp/kt_mono.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt_mono -name "*results.db" | head -1) 1 | grep -v "at::native\|Cijk\|rocclr" | head -20
