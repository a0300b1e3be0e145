cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
python tools/debug/emulate_ranks.py --worlds 4,8 --steps 10 > $O/emu_dbg.txt 2>&1; tail -25 $O/emu_dbg.txt | cut -c1-300
