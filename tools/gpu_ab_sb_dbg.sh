# timing ablations of sad_strip_kernel (AOMHIP_SB_DBG bits: 1 no evaluation, 2 list slices only, 32 decode only, 128 no reduction, 64 barriers only)
for D in ${DBGS:-0 2 34 130 3 64}; do echo "dbg=$D"; AOMHIP_SB_DBG=$D timeout 300 python tools/gpu_ab_sadsb.py ${ARGS:-4k 8 64 480,32} 2>&1 | grep '^{' | cut -c1-90; done
