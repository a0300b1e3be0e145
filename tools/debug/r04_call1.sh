cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(time timeout 2400 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "sharded or rccl or two_ranks or thread_safe or serving_loop or mono_single or c5_geometry_two" -p no:cacheprovider) > gpurun_out/r04/t1.log 2>&1
tail -15 gpurun_out/r04/t1.log
(time timeout 900 python tools/debug/emulate_ranks.py --worlds 1,2,4,8 --steps 10) > gpurun_out/r04/emu_c3.log 2>&1
tail -12 gpurun_out/r04/emu_c3.log | cut -c1-400
(time timeout 600 python bench.py) > gpurun_out/r04/bench1.json 2> gpurun_out/r04/bench1.err
cut -c1-600 gpurun_out/r04/bench1.json
