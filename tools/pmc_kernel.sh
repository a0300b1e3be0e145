#!/bin/bash
# Generic PMC passes for kernels matching a regex (separate passes; counters only, no trace domains).
# usage: tools/pmc_kernel.sh <kernel-regex> <outdir-name> [bench args...]
set -u
cd /tmp && export TMPDIR=/tmp
RX="$1"; NAME="$2"; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-include-regex "$RX" --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --reps 1 --no-cpu-baseline --no-c5 "$@" > $OUT/p$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen and r["Counter_Name"] in ("SQ_WAVES", "SQ_WAIT_INST_ANY", "SQ_INSTS_SMEM"):
            seen.add(key)
    for k, d in {(k): None for k, _ in seen}.items():
        pass
for k, d in agg.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:.4g}")
PY
