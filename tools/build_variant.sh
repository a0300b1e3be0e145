#!/bin/bash
# variants_<name>.so at the repo root = the library with ONE source rebuilt with extra -D switches (A/B builds; git-ignored)
#   tools/build_variant.sh <name> <source stem, e.g. hhsr_merge> "<-D flags>"
set -e
P=handheld-multi-frame-super-resolution_amd
NAME=$1; SRC=$2; FLAGS=$3
cd "$(dirname "$0")/.."
python $P/build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -fno-slp-vectorize \
  $FLAGS -c $P/csrc/$SRC.hip -o /tmp/variant_${NAME}_$SRC.o
OBJS=$(ls $P/build/hhsr_*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants_$NAME.so $OBJS /tmp/variant_${NAME}_$SRC.o -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
echo variants_$NAME.so
