#!/bin/bash
# round 6: experiment builds of csrc/sad_sb.hip (16x16 instantiations only) for same-box A/B runs.
#   bash tools/sb_build_exp.sh name "flags" [name "flags" ...]   ->  explib/libsadsb_<name>.so and explib/libsadsb_<name>_prof.so
# Each is ONLY aomhip_sad_sb_batch (+ the phase-clock read-out) linked against the product library for the rest; the A/B tools rebind that
# one entry point to it (tools/sb_override.py, AOMHIP_SB_LIB=...).  explib/ travels to the GPU box; build/ does not.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iaom-av1-psy_amd/csrc -Wall -Wno-unused-function -DAOMHIP_SB_ONLY_16 -DAOMHIP_SB_DBG_KNOBS=1"
LINK="-Laom-av1-psy_amd/lib -laomhip -Wl,-rpath,\$ORIGIN/../aom-av1-psy_amd/lib"
mkdir -p explib build/exp
while [ $# -ge 2 ]; do
  NAME=$1; EXTRA=$2; shift 2
  ( $HIPCC $FLAGS $EXTRA -c aom-av1-psy_amd/csrc/sad_sb.hip -o build/exp/sad_sb_$NAME.o 2>&1 | grep -v "warning generated" || true
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o explib/libsadsb_$NAME.so build/exp/sad_sb_$NAME.o $LINK ) &
  ( $HIPCC $FLAGS $EXTRA -DAOMHIP_SB_PROF -c aom-av1-psy_amd/csrc/sad_sb.hip -o build/exp/sad_sb_${NAME}_prof.o 2>&1 | grep -v "warning generated" || true
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o explib/libsadsb_${NAME}_prof.so build/exp/sad_sb_${NAME}_prof.o $LINK ) &
  wait
done
ls -la explib/
