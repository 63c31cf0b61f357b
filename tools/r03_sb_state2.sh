mkdir -p gpurun_out/r03c
export AOMHIP_LIB=build/exp/libaomhip_exp.so
for W in "4k 8 64 320,48"; do
  for D in 0 2 514 1026 1538 512 1024; do echo "args=$W dbg=$D"; AOMHIP_SB_DBG=$D timeout 300 python tools/gpu_ab_sadsb.py $W 2>&1 | grep '^{"cell' | cut -c1-120; done
done > gpurun_out/r03c/ablate.log 2>&1
cat gpurun_out/r03c/ablate.log
