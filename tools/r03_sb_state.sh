# round 3: where the strip kernel stands (ablations + phase clocks), 16x16 experiment builds (make exp)
mkdir -p gpurun_out/r03b
export AOMHIP_LIB=build/exp/libaomhip_exp.so
for W in "4k 8 64 320,48" "1080p 8 64 240,64" "4k 10 32 160,32"; do
  for D in 0 1 2 34 130 3 64; do echo "args=$W dbg=$D"; AOMHIP_SB_DBG=$D timeout 300 python tools/gpu_ab_sadsb.py $W 2>&1 | grep '^{"cell' | cut -c1-120; done
done > gpurun_out/r03b/ablate.log 2>&1
AOMHIP_LIB=build/exp/libaomhip_exp_prof.so python tools/gpu_sb_prof.py 4k 8 64 320,48 > gpurun_out/r03b/prof.log 2>&1
AOMHIP_SB_DBG=2 AOMHIP_LIB=build/exp/libaomhip_exp_prof.so python tools/gpu_sb_prof.py 4k 8 64 320,48 >> gpurun_out/r03b/prof.log 2>&1
tail -30 gpurun_out/r03b/ablate.log; tail -3 gpurun_out/r03b/prof.log
