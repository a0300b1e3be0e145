#!/bin/bash
# Everything the round's profiles / PARITY.md are made from, in one GPU call.  usage: tools/collect_round.sh <name>
NAME=${1:-r02}
OUT=gpurun_out/$NAME
mkdir -p $OUT
HHSR_PARITY_LOG=$PWD/$OUT/parity.jsonl timeout 2400 python -m pytest tests -q -m gpu --timeout 1200 > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
python tools/parity_report.py $OUT/parity.jsonl > $OUT/PARITY.md
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python bench.py --height 6000 --width 8000 --scale 3 --frames 20 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c5.json 2> /dev/null
python bench.py --frames 8 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c2.json 2> /dev/null
bash tools/kernel_trace.sh $NAME/kt 5 > /dev/null 2>&1
bash tools/pmc_all.sh $NAME/pmc_all > /dev/null 2>&1
bash tools/pmc_merge.sh "k_merge_x2" $NAME/pmc_x2 --no-h2d --steps 1 --warmup 0 > /dev/null 2>&1
python tools/pmc_report.py $OUT/pmc_x2 k_merge_x2 > $OUT/pmc_x2.md
tools/ubench/valu_rate > $OUT/valu_rate.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /tmp/ldsp -o lds -- $GRAFT_REPO_ROOT/tools/ubench/lds_patterns > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/ubench/pmc_lds.py /tmp/ldsp > $GRAFT_REPO_ROOT/$OUT/lds_patterns.txt)
find $OUT -name "*agent_info*" -delete
cut -c1-600 $OUT/bench_n1.json
