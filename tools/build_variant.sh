#!/bin/bash
# variants_<name>.so at the repo root = the library with ONE source rebuilt with extra -D switches (A/B builds; git-ignored)
#   tools/build_variant.sh <name> <source stem, e.g. hhsr_merge> "<-D flags>"
set -e
P=handheld-multi-frame-super-resolution_amd
NAME=$1; SRC=$2; FLAGS=$3
cd "$(dirname "$0")/.."
python $P/build.py > /dev/null
# (the source's own flags from build.py: the merge sources keep hipcc's default contraction, the others build with it off)
OWN=$(python -c "import importlib.util as u; s=u.spec_from_file_location('b','$P/build.py'); m=u.module_from_spec(s); s.loader.exec_module(m); print(' '.join(m.SOURCES['$SRC.hip']))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $OWN \
  $FLAGS -c $P/csrc/$SRC.hip -o /tmp/variant_${NAME}_$SRC.o
OBJS=$(ls $P/build/hhsr_*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants_$NAME.so $OBJS /tmp/variant_${NAME}_$SRC.o -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
echo variants_$NAME.so
