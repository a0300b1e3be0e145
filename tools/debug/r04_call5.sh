cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python tools/debug/border_cost.py > gpurun_out/r04/x3_border_cost.txt 2>&1; cat gpurun_out/r04/x3_border_cost.txt
bash tools/debug/ab_c5.sh xsbase xsilp > gpurun_out/r04/ab_c5_c.txt 2>&1; cat gpurun_out/r04/ab_c5_c.txt
(time timeout 3000 python -m pytest tests/ -q -m gpu -p no:cacheprovider -x) > gpurun_out/r04/full1.log 2>&1
tail -15 gpurun_out/r04/full1.log
