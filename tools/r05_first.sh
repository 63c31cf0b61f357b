# round 5, first GPU call: (1) the bench line's size and schema, (2) the default bench, (3) A/B of sad_sb.hip at b18cb84 (round 3's profile)
# vs HEAD (two __umul24 address products) on one box, (4) PMC traffic passes refreshed for the SAD rings and the transform kernels
set -u
export TMPDIR=/tmp
T=${1:-r05a}
OUT=gpurun_out/$T; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_bench_schema.py -x -q > $OUT/schema.log 2>&1; tail -3 $OUT/schema.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; wc -c $OUT/bench_default.json
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
OUT=$T/ab REPS=3 AB_REPS=30 LIBS="explib/libaomhip_exp_b18.so explib/libaomhip_exp_head.so" WORK="4k 8 64 320,48;1080p 8 64 240,64;4k 10 32 160,32" bash tools/r03_ab.sh > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
bash tools/gpu_pmc_sb.sh $T > $OUT/pmc_sb.log 2>&1
bash tools/gpu_pmc_txq.sh $T txq_1080p_8bit > $OUT/pmc_txq.log 2>&1
bash tools/gpu_pmc_txq.sh $T txq_4k_10bit >> $OUT/pmc_txq.log 2>&1
cp profiles/traffic.json profiles/${T}_pmc_* $OUT/ 2>/dev/null
cat profiles/traffic.json
