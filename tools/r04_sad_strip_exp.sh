# round 4, verdict item 4: the three bounded sad_strip_kernel experiments on one box (experiment builds of csrc/sad_sb.hip, 16x16 only).
#   (a) results through an LDS tile streamed out by the loader wavefronts (-DAOMHIP_SB_LDS_OUT=1) vs the shipped form
#   (b) upper bound of a grid-mode entry point: AOMHIP_SB_DBG=8192 (no list slices travel, entries are a function of their index; results invalid)
#   (c) a range-32 contract (lists within +-32, LDS halo 32) with the cells that then fit, next to range 64
export AB_REPS=30
OUT=gpurun_out/r04_sad_exp; mkdir -p $OUT
OUT=r04_sad_exp/a REPS=3 LIBS="build/exp/libaomhip_exp_base.so build/exp/libaomhip_exp_ldsout.so" bash tools/r03_ab.sh > $OUT/a.txt 2>&1
OUT=r04_sad_exp/b REPS=2 LIBS="build/exp/libaomhip_exp_base.so" DBGS="0 8192" bash tools/r03_ab.sh > $OUT/b.txt 2>&1
for R in 64 32; do
  AB_RANGE=$R OUT=r04_sad_exp/c10_$R REPS=2 LIBS="build/exp/libaomhip_exp_base.so" WORK="4k 10 32 160,32 224,32 256,32 160,48 320,16" bash tools/r03_ab.sh > $OUT/c10_$R.txt 2>&1
  AB_RANGE=$R OUT=r04_sad_exp/c8_$R REPS=2 LIBS="build/exp/libaomhip_exp_base.so" WORK="4k 8 64 320,48 480,32 384,32 240,64;1080p 8 64 240,64 480,32 320,48" bash tools/r03_ab.sh > $OUT/c8_$R.txt 2>&1
done
tail -n +1 $OUT/*.txt
