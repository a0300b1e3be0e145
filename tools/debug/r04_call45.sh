cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "rccl" -p no:cacheprovider > $O/rccl_dbg.txt 2>&1; grep -v "^$" $O/rccl_dbg.txt | tail -60 | cut -c1-250
