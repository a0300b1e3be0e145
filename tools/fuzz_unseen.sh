#!/bin/bash
# More generator seeds nobody has run, with the frozen rules (report mode): tools/fuzz_unseen.sh <commit> <out> seed...
COMMIT=$1; OUT=$2; shift 2
mkdir -p "$(dirname "$OUT")"
RULES=$(cat tests/helpers.py tests/test_fuzz_parity.py | sha256sum | cut -c1-16)
B=""
for g in "$@"; do B="$B${B:+,}$g:22,$((g+1)):22,$((g+2)):20"; done
echo "# frozen fuzz contract on unseen generator seeds $*: commit $COMMIT, rules sha256[:16] $RULES" > "$OUT"
HHSR_FUZZ_BATCHES="$B" HHSR_FUZZ_REPORT="$OUT" python -m pytest tests/test_fuzz_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
echo "# cases: $(grep -c '^case' "$OUT"), cases violating an assertion: $(grep -c 'ASSERTIONS FAILED' "$OUT")" >> "$OUT"
tail -1 "$OUT"
