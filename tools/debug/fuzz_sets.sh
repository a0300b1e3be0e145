# the fuzz sweep in report mode on the 64 fixed cases and four held-out sets of 64 (other generator seeds)
mkdir -p gpurun_out/fuzz
i=0
for B in "" "10:22,11:22,12:20" "20:22,21:22,22:20" "30:22,31:22,32:20" "40:22,41:22,42:20"; do
  rm -f gpurun_out/fuzz/set$i.txt
  HHSR_FUZZ_BATCHES="$B" HHSR_FUZZ_REPORT=$PWD/gpurun_out/fuzz/set$i.txt python -m pytest tests/test_fuzz_parity.py -m gpu -q 2>&1 | tail -1
  grep -c "ASSERTIONS FAILED" gpurun_out/fuzz/set$i.txt
  i=$((i+1))
done
