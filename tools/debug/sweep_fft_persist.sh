# resident row workgroups of the FFT kernels (3 per CU fill the LDS: 768) vs the graph-replayed step, where other
# streams' kernels could use the LDS a smaller FFT grid leaves free
for p in 768 512 640 384 1024; do
  HHSR_FFT_PERSIST=$p python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist $p: graph', d['ms_per_step'], 'eager', d['ms_per_step_eager'])"
done
