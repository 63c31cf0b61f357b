set -u; export TMPDIR=/tmp
T=${1:-r05c}; OUT=gpurun_out/$T; mkdir -p $OUT
python tools/r05_valu_issue.py $T > $OUT/valu_issue.md 2>&1
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*" | sort -u > $OUT/valu_counters.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/il -o il -- python3 bench.py --workload inner_loop_4k_10bit --steps 60 --warmup 3 > $OUT/il_prof.json 2> $OUT/il_prof.err
python3 tools/r05_timeline.py $OUT/il 20 > $OUT/timeline.md 2>&1; cat $OUT/timeline.md
AOMHIP_BENCH_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/il_nograph -o il -- python3 bench.py --workload inner_loop_4k_10bit --steps 60 --warmup 3 > $OUT/il_nograph.json 2> $OUT/il_nograph.err
python3 tools/r05_timeline.py $OUT/il_nograph 20 > $OUT/timeline_nograph.md 2>&1; cat $OUT/timeline_nograph.md
python3 bench.py --workload inner_loop_4k_10bit --steps 60 --warmup 3 > $OUT/il.json 2> $OUT/il.err; cut -c1-3000 $OUT/il.json
rm -rf $OUT/il/*/*agent_info.csv
cat $OUT/valu_counters.txt | tr '\n' ' '
