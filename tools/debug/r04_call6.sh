cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "x3 or c5_geometry or merge_golden or e2e_golden_scales or align or bm_ or ica" -p no:cacheprovider) > gpurun_out/r04/t6.log 2>&1
tail -6 gpurun_out/r04/t6.log
python tools/debug/border_cost.py > gpurun_out/r04/x3_border_cost_b.txt 2>&1; cat gpurun_out/r04/x3_border_cost_b.txt
bash tools/debug/ab_c5.sh default > gpurun_out/r04/ab_c5_d.txt 2>&1; cat gpurun_out/r04/ab_c5_d.txt
(time timeout 600 python bench.py --no-cpu-baseline --no-h2d) > gpurun_out/r04/bench2.json 2> gpurun_out/r04/bench2.err
cut -c1-300 gpurun_out/r04/bench2.json
(time timeout 3000 python -m pytest tests/ -q -m gpu -p no:cacheprovider) > gpurun_out/r04/full2.log 2>&1
tail -12 gpurun_out/r04/full2.log
