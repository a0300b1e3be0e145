#!/bin/bash
# ONE entry point for the A/B measurements of library variants (replaces round 4's per-call scratch scripts).
# A variant = variants_<name>.so at the repo root (tools/build_variant.sh <name> <source stem> "<-D flags>") and / or
# environment switches of the library.  Run on an MI355X (gpurun), from the repo root.
#
#   tools/ab.sh [options] CONFIG [CONFIG ...]
#     CONFIG   label[@variant][,ENV=VALUE ...]     "default" = the committed library; e.g.  pf5@pf5  nt256@nt256,HHSR_FFT_PERSIST=1280
#   options
#     --kernels REGEX     rocprofv3 --kernel-trace of an eager one-stream bench run per config; prints the per-kernel rows
#                         matching REGEX (plus the total)
#     --bench N           N alternating rounds of the graph-replayed bench per config: ms_per_step / eager / merge launch ms
#     --tests KEXPR       first run `pytest tests/test_hip_parity.py -m gpu -k KEXPR` against every variant library
#     --size "H W F S"    burst geometry for --kernels / --bench (default: the headline 3000 4000 20 2; C5: "6000 8000 20 3")
#     --steps N           steps of the traced run (default 5)
# Examples (what round 4 ran as tools/debug/r04_call*.sh):
#   tools/ab.sh --kernels "k_rows|k_cols" --tests "grey or fft" default pf@pf,HHSR_FFT_PERSIST=512      (FFT prefetch variants)
#   tools/ab.sh --bench 3 default serial,HHSR_MERGE_BORDER_SERIAL=1                                     (border bands)
#   tools/ab.sh --bench 2 --size "6000 8000 20 3" default nw2@nw2                                       (x3 merge variants)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
KERNELS=""; BENCH=0; TESTS=""; SIZE="3000 4000 20 2"; STEPS=5
while [ $# -gt 0 ]; do
  case "$1" in
    --kernels) KERNELS="$2"; shift 2;;
    --bench) BENCH="$2"; shift 2;;
    --tests) TESTS="$2"; shift 2;;
    --size) SIZE="$2"; shift 2;;
    --steps) STEPS="$2"; shift 2;;
    *) break;;
  esac
done
read -r H W F S <<< "$SIZE"
GEOM="--height $H --width $W --frames $F --scale $S"
envof() {  # CONFIG -> "HHSR_LIB=... K=V ..." on stdout, label in $LABEL
  local cfg="$1" head rest lib=""
  head="${cfg%%,*}"; rest=""; [ "$cfg" != "$head" ] && rest="${cfg#*,}"
  LABEL="${head%%@*}"
  [ "$head" != "$LABEL" ] && lib="$ROOT/variants_${head#*@}.so"
  echo "HHSR_LIB=$lib ${rest//,/ }"
}
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  E=$(envof "$cfg"); LABEL="${cfg%%[@,]*}"
  if [ -n "$TESTS" ]; then
    echo "== tests [$LABEL]"; (cd "$ROOT" && env $E timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -k "$TESTS" -p no:cacheprovider 2>&1 | tail -2)
  fi
  if [ -n "$KERNELS" ]; then
    rm -rf "/tmp/kt_$LABEL"
    env $E rocprofv3 --kernel-trace --stats -d "/tmp/kt_$LABEL" -o kt -- python "$ROOT/bench.py" --reps 1 --no-cpu-baseline --no-h2d --no-graph \
      --steps "$STEPS" --warmup 2 --streams 1 $GEOM > "/tmp/kt_$LABEL.log" 2>&1
    echo "== kernels [$LABEL] ($E)"
    python "$ROOT/tools/rocprof_summary.py" "$(find "/tmp/kt_$LABEL" -name '*results.db' | head -1)" $((STEPS + 2)) | grep -E "$KERNELS|Total" | cut -c1-160
  fi
done
for ((i = 1; i <= BENCH; i++)); do
  for cfg in "$@"; do
    E=$(envof "$cfg"); LABEL="${cfg%%[@,]*}"
    env $E python "$ROOT/bench.py" --no-cpu-baseline --no-h2d --steps 15 --warmup 4 $GEOM 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print('$LABEL step', d['ms_per_step'], 'eager', d['ms_per_step_eager'], r.get('kernel'), r.get('avg_launch_ms'))"
  done
done
