#!/bin/bash
# Which kernels of the library use scratch (private segment) / how many registers: compiles every csrc/*.hip to device assembly (no GPU
# needed, ~2.5 min on 8 cores) and reads the kernel metadata.  Usage: bash tools/list_scratch_kernels.sh [outdir]
OUT=${1:-/tmp/aomhip_asm}
mkdir -p $OUT
ls aom-av1-psy_amd/csrc/*.hip | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iaom-av1-psy_amd/csrc -w --offload-device-only -S {} -o $OUT/\$(basename {} .hip).s 2>/dev/null"
python3 - "$OUT" <<'PY'
import re, glob, sys
tot, sc = 0, []
for f in glob.glob(sys.argv[1] + '/*.s'):
    t = open(f).read()
    if 'amdhsa.kernels' not in t:
        continue
    k = t[t.index('amdhsa.kernels'):]
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.vgpr_spill_count:\s+(\d+)", k, re.S):
        if m.group(1).endswith('.kd'):
            continue
        ps = re.search(r"\.private_segment_fixed_size:\s+(\d+)", m.group(2))
        vg = re.search(r"\.vgpr_count:\s+(\d+)", m.group(2))
        tot += 1
        if ps and int(ps.group(1)) > 0:
            sc.append((int(ps.group(1)), int(vg.group(1)) if vg else -1, int(m.group(3)), m.group(1)))
print(tot, 'kernels;', len(sc), 'with scratch (bytes, VGPRs, spilled VGPRs, name):')
for s in sorted(sc, reverse=True):
    print(*s[:3], s[3][:160])
PY
