# the fuzz sweep in report mode on further held-out generator seeds: tools/debug/fuzz_more_sets.sh 50 60 70 80 90
mkdir -p gpurun_out/fuzz
for g in "$@"; do
  rm -f gpurun_out/fuzz/set$g.txt
  HHSR_FUZZ_BATCHES="$g:22,$((g+1)):22,$((g+2)):20" HHSR_FUZZ_REPORT=$PWD/gpurun_out/fuzz/set$g.txt python -m pytest tests/test_fuzz_parity.py -m gpu -q 2>&1 | tail -1
  grep "ASSERTIONS FAILED" gpurun_out/fuzz/set$g.txt | cut -c1-700
done
