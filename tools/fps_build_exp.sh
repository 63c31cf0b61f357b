#!/bin/bash
# Experiment builds of the full-pel search kernels (csrc/fullpel_search.inc, 16-bit planes, 16x16 blocks only) for same-box A/B runs.
#   bash tools/fps_build_exp.sh name "flags" [name "flags" ...]   ->  explib/libfps_<name>.so and explib/libfps_<name>_prof.so
# Each holds aomhip_full_pixel_search_batch with its own launch_fps_u16 (+ the phase-clock read-out in the _prof form) linked against the product
# library for the rest; tools/fps_ab.py rebinds that one entry point of the ctypes binding to it (the product's internal calls -- the temporal
# filter, the first pass -- stay the product's: the library binds them at link time).  explib/ travels to the GPU box; build/ does not.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iaom-av1-psy_amd/csrc -Wall -Wno-unused-function -DAOMHIP_FPS_ONLY_16"
LINK="-Laom-av1-psy_amd/lib -laomhip -Wl,-rpath,\$ORIGIN/../aom-av1-psy_amd/lib"
mkdir -p explib build/exp
while [ $# -ge 2 ]; do
  NAME=$1; EXTRA=$2; shift 2
  [ -f build/exp/fps_entry.o ] || $HIPCC $FLAGS -c aom-av1-psy_amd/csrc/fullpel_search.hip -o build/exp/fps_entry.o
  ( $HIPCC $FLAGS $EXTRA -c aom-av1-psy_amd/csrc/fullpel_search_u16.hip -o build/exp/fps_$NAME.o 2>&1 | grep -v "warning generated" || true
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o explib/libfps_$NAME.so build/exp/fps_entry.o build/exp/fps_$NAME.o $LINK ) &
  ( $HIPCC $FLAGS $EXTRA -DAOMHIP_FPS_PROF -c aom-av1-psy_amd/csrc/fullpel_search_u16.hip -o build/exp/fps_${NAME}_prof.o 2>&1 | grep -v "warning generated" || true
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o explib/libfps_${NAME}_prof.so build/exp/fps_entry.o build/exp/fps_${NAME}_prof.o $LINK ) &
  wait
done
ls -la explib/
