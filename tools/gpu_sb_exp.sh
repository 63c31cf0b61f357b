# A/B + phase profile of the experiment builds of sad_strip_kernel (make exp): usage  bash tools/gpu_sb_exp.sh [dbg values for the profile]
export AOMHIP_SB_CFG=${CFG:-wide}
export AOMHIP_LIB=build/exp/libaomhip_exp.so
python tools/gpu_ab_sadsb.py 4k 8 64 ${CELLS:-240,48} 2>&1 | grep '^{"cell' | cut -c1-200
python tools/gpu_ab_sadsb.py 1080p 8 64 ${CELLS1080:-240,48} 2>&1 | grep '^{"cell' | cut -c1-200
python tools/gpu_ab_sadsb.py 4k 10 32 160,32 2>&1 | grep '^{"cell' | cut -c1-200
for D in ${@:-0}; do echo "prof dbg=$D"; AOMHIP_SB_DBG=$D AOMHIP_LIB=build/exp/libaomhip_exp_prof.so python tools/gpu_sb_prof.py 4k 8 64 ${PCELL:-240,48} 2>&1 | tail -1; done
