#!/bin/bash
# Experiment builds of csrc/mcomp.hip (the diamond / mesh / bilinear sub-pel kernels) for same-box A/B runs: explib/libmcomp_<name>.so holds the file's entry
# points linked against the product library for the rest; tools/fps_ab.py --workloads diamond rebinds aomhip_fullpel_diamond_batch to it.
#   bash tools/mcomp_build_exp.sh name "flags" [name "flags" ...]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iaom-av1-psy_amd/csrc -Wall -Wno-unused-function"
LINK="-Laom-av1-psy_amd/lib -laomhip -Wl,-rpath,\$ORIGIN/../aom-av1-psy_amd/lib"
mkdir -p explib build/exp
while [ $# -ge 2 ]; do
  NAME=$1; EXTRA=$2; shift 2
  ( $HIPCC $FLAGS $EXTRA -c aom-av1-psy_amd/csrc/mcomp.hip -o build/exp/mcomp_$NAME.o 2>&1 | grep -v "warning generated" || true
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o explib/libmcomp_$NAME.so build/exp/mcomp_$NAME.o $LINK ) &
done
wait
ls -la explib/ | grep mcomp
