# is sad_strip_kernel clock / power limited?  the same launch on random, smooth and all-zero frames + the phase clocks (cycles) next to the wall time
mkdir -p gpurun_out/r03j
for D in lcg smooth zero; do for R in 1 2; do echo "data=$D"; AB_DATA=$D AB_REPS=30 AOMHIP_LIB=build/exp_base/libaomhip_exp.so python tools/gpu_ab_sadsb.py 4k 8 64 320,48 2>&1 | grep '^{"cell' | cut -c1-200; done; done 2>&1 | tee gpurun_out/r03j/power.log
for D in lcg zero; do echo "prof data=$D"; AB_DATA=$D AOMHIP_LIB=build/exp/libaomhip_exp_prof.so python tools/gpu_sb_prof.py 4k 8 64 320,48 2>&1 | tail -1; done 2>&1 | tee -a gpurun_out/r03j/power.log
(while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" ; sleep 0.5; done) > gpurun_out/r03j/smi.log 2>&1 &
SMI=$!
AB_REPS=20000 AOMHIP_LIB=build/exp_base/libaomhip_exp.so python tools/gpu_ab_sadsb.py 4k 8 64 320,48 2>&1 | grep '^{"cell' | cut -c1-200 | tee -a gpurun_out/r03j/power.log
kill $SMI
tail -12 gpurun_out/r03j/smi.log
