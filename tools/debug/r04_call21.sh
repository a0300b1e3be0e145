cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
bash tools/debug/ab_c5.sh xsbase2 xspipe0 xsocc3 > gpurun_out/r04/ab_c5_f.txt 2>&1; cat gpurun_out/r04/ab_c5_f.txt
