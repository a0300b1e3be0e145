cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
bash tools/debug/kt_ab.sh "k_merge_x2|k_align_wave|k_gauss|k_rows|k_cols|Total" slpmerge slpalign slppyramid > gpurun_out/r04/kt_ab_slp.txt 2>&1; cat gpurun_out/r04/kt_ab_slp.txt
(timeout 600 python bench.py --no-cpu-baseline --no-h2d) > gpurun_out/r04/bench3.json 2>/dev/null; cut -c1-260 gpurun_out/r04/bench3.json
