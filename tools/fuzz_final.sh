#!/bin/bash
# The two-sided fuzz contract (tests/helpers.py: fuzz_verdict + constants; tests/test_fuzz_parity.py) on every set in ONE run
# of ONE commit, as ONE sweep: the 64 fixed cases, every set earlier rounds ran (generator seeds 10 .. 110, 200, 300, and the
# 14 "unseen" seeds of round 4, the 8 seeds 2000 .. 2700 that round 5's first two-sided run saw) and the seeds given on the
# command line (never run before).
#   tools/fuzz_final.sh <commit-hash> <out-file> [new seed ...]        (on an MI355X; the oracle pairs run on the host cores)
COMMIT=${1:-unknown}
OUT=${2:-$PWD/gpurun_out/r05_fuzz_final.txt}
shift 2
NEW="$*"
OLD="10 20 30 40 50 60 70 80 90 100 110 200 300 400 500 600 700 800 900 1000 1100 1200 1300 1400 1500 1600 1700 2000 2100 2200 2300 2400 2500 2600 2700 3000 3100 3200 3300 4000 4100 4200 4300 4400 4500 4600 4700 5000 5100 5200 5300 6000 6100 6200 6300 6400 6500 6600 6700"
[ -n "$HHSR_FUZZ_OLD" ] && OLD="$HHSR_FUZZ_OLD"      # (a subset, for a trial run)
mkdir -p "$(dirname "$OUT")"
RULES=$(cat tests/helpers.py tests/test_fuzz_parity.py | sha256sum | cut -c1-16)
B="0:22,1:22,2:20"
for g in $OLD $NEW; do B="$B,$g:22,$((g+1)):22,$((g+2)):20"; done
{
  echo "# two-sided fuzz contract: commit $COMMIT, rules sha256[:16] (tests/helpers.py + tests/test_fuzz_parity.py) $RULES"
  echo "# sets: fixed 0-2 | run in earlier rounds: $OLD | NEW, never run before: ${NEW:-none}"
  echo "# constants: $(grep -E '^(FLIP_PX|MAX_ICA_TILES|CLUSTER|MAX_FLIP_TILES|ACC_TOL) =' tests/helpers.py | sed 's/ *#.*//' | tr '\n' ';')"
  echo "# per case: alignment (flows) | side H: HIP own flows vs ORACLE ON HIP'S FLOWS (whole chain; merge alone on HIP's flows + HIP's r) | side O: the same on the oracle's flows | informational"
} > "$OUT"
T0=$(date +%s)
HHSR_FUZZ_BATCHES="$B" HHSR_FUZZ_REPORT="$OUT" python -m pytest tests/test_fuzz_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
echo "# cases: $(grep -c '^case' "$OUT"), cases violating an assertion: $(grep -c 'ASSERTIONS FAILED' "$OUT"), wall $(( $(date +%s) - T0 )) s" >> "$OUT"
tail -1 "$OUT"
