cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
HHSR_DBG_NOGC=1 timeout 600 python tools/debug/rccl_worker_dbg.py > $O/rccl_dbg3.txt 2>&1; echo "NOGC rc=$?"; tail -4 $O/rccl_dbg3.txt | cut -c1-200
timeout 600 python tools/debug/rccl_worker_dbg.py > $O/rccl_dbg2.txt 2>&1; echo "GC-held rc=$?"; tail -4 $O/rccl_dbg2.txt | cut -c1-200
