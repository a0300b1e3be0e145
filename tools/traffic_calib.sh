#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on the merge kernel's access patterns -> gpurun_out/<name>/traffic_calib.md
NAME=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tc_f /tmp/tc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/tc_f -o f -- $GRAFT_REPO_ROOT/tools/ubench/traffic_calib > $OUT/traffic_calib.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/tc_w -o w -- $GRAFT_REPO_ROOT/tools/ubench/traffic_calib >> $OUT/traffic_calib.log 2>&1
python $GRAFT_REPO_ROOT/tools/ubench/traffic_calib_report.py /tmp/tc_f /tmp/tc_w > $OUT/traffic_calib.md
cat $OUT/traffic_calib.md
