#!/bin/bash
# A whole library whose 16-bit sub-pel kernels carry the phase clocks of the up-sampled error (-DAOMHIP_UPS_PROF, csrc/subpel_search.inc):
#   bash tools/ups_build_prof.sh [extra flags]   ->  explib/libaomhip_upsprof.so   (AOMHIP_LIB=explib/libaomhip_upsprof.so python tools/ups_prof.py)
# The product's objects (build/*.o, `make lib` first) with subpel_search_u16.o replaced: the temporal filter's and the compound searches' internal calls reach it.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p explib build/exp
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iaom-av1-psy_amd/csrc -Wall -Wno-unused-function -DAOMHIP_UPS_PROF "$@" -c aom-av1-psy_amd/csrc/subpel_search_u16.hip -o build/exp/subpel_u16_prof.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o explib/libaomhip_upsprof.so $(ls build/*.o | grep -v "build/subpel_search_u16.o") build/exp/subpel_u16_prof.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
ls -la explib/libaomhip_upsprof.so
