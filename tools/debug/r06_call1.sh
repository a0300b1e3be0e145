set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
python bench.py > gpurun_out/r06a/bench_n1.json 2> gpurun_out/r06a/bench_n1.err
tail -c 3000 gpurun_out/r06a/bench_n1.err
bash tools/kernel_trace.sh r06a/kt 5 
