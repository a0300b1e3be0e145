#!/bin/bash
# The FROZEN fuzz contract (tests/helpers.py: fuzz_verdict + constants) on every set in ONE run of ONE commit:
# the 64 fixed cases, round 3's eleven held-out sets (generator seeds 10.. 110) and two seeds nobody had run (200, 300).
#   tools/fuzz_final.sh <commit-hash> [out-file]        (on an MI355X; ~2 min per set of 64: the oracle runs on the host)
COMMIT=${1:-unknown}
OUT=${2:-$PWD/gpurun_out/r04_fuzz_final.txt}
mkdir -p "$(dirname "$OUT")"
RULES=$(cat tests/helpers.py tests/test_fuzz_parity.py | sha256sum | cut -c1-16)
B="0:22,1:22,2:20"
for g in 10 20 30 40 50 60 70 80 90 100 110 200 300; do B="$B,$g:22,$((g+1)):22,$((g+2)):20"; done
{
  echo "# fuzz contract, final run: commit $COMMIT, rules sha256[:16] (tests/helpers.py + tests/test_fuzz_parity.py) $RULES"
  echo "# sets: fixed 0-2 | held-out (round 3) 10 20 30 40 50 60 70 80 90 100 110 | NEW, never run before: 200 300"
  echo "# constants: $(grep -E '^(FLIP_PX|MAX_ICA_TILES|CLUSTER|MAX_OUTLIERS|MAX_OUTLIER|DEN_FLOOR|NUM_ERR|MAX_SENS|MAX_FLIP_TILES) =' tests/helpers.py | sed 's/ *#.*//' | tr '\n' ';')"
} > "$OUT"
HHSR_FUZZ_BATCHES="$B" HHSR_FUZZ_REPORT="$OUT" python -m pytest tests/test_fuzz_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
echo "# cases: $(grep -c '^case' "$OUT"), cases violating an assertion: $(grep -c 'ASSERTIONS FAILED' "$OUT")" >> "$OUT"
tail -1 "$OUT"
