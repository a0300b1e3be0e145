cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
python tools/debug/emulate_ranks.py --worlds 8 --steps 10 --strategies reduce 2>&1 | tail -3 | cut -c1-300
python tools/debug/emulate_ranks.py --worlds 8 --steps 3 --strategies reduce --height 6000 --width 8000 --scale 3 2>&1 | tail -3 | cut -c1-300
