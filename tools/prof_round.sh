#!/bin/bash
# usage: tools/prof_round.sh <rNN> [a|b]  (a: bench lines, sweep, kernel stats + SQ counters; b: PMC traffic + back-half stall counters; default both) -- everything profiles/ holds for a round, under gpurun_out/<rNN>_*: bench lines, rocprofv3 kernel stats,
# SQ counters and PMC traffic of the dominant kernels (cfg2 full, cfg2 / cfg3 125-group shares = what one of 8 ranks scores, cfg4, cfg5)
R=$1
PART=${2:-ab}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
if [[ $PART == *a* ]]; then
B="timeout -k 10 300 python3 bench.py"
$B --steps 20 --warmup 3 > $OUT/${R}_cfg2_bench.json 2> $OUT/${R}_cfg2_bench.err
$B --groups 125 --steps 20 --warmup 3 --e2e 0 --cpu-groups 0 > $OUT/${R}_cfg2_125groups_bench.json 2> /dev/null
$B --config cfg3 --groups 125 --steps 10 --warmup 2 --e2e 0 --cpu-groups 0 > $OUT/${R}_cfg3_125groups_bench.json 2> /dev/null
$B --config cfg4 --steps 5 --warmup 1 > $OUT/${R}_cfg4_bench.json 2> $OUT/${R}_cfg4_bench.err
$B --config cfg5 --steps 10 --warmup 2 --e2e 0 --cpu-groups 0 > $OUT/${R}_cfg5_bench.json 2> /dev/null
$B --output group --steps 10 --warmup 2 --e2e 0 --cpu-groups 0 > $OUT/${R}_cfg2_group_bench.json 2> /dev/null
bash tools/sweep_groups.sh 2 8 32 125 250 500 1000 > $OUT/${R}_cfg2_sweep_groups.txt 2>&1
bash tools/prof_sq.sh ${R}_cfg2 > $OUT/${R}_cfg2_prof.log 2>&1
bash tools/prof_sq.sh ${R}_cfg3_share --config cfg3 --groups 125 > $OUT/${R}_cfg3_prof.log 2>&1
bash tools/prof_sq.sh ${R}_cfg4 --config cfg4 > $OUT/${R}_cfg4_prof.log 2>&1
fi
if [[ $PART == *b* ]]; then
bash tools/prof_traffic.sh ${R}_cfg2 cfg2 "score_quad_kernel reduce_buckets_kernel km_write_lines_kernel" > $OUT/${R}_cfg2_traffic.log 2>&1
bash tools/prof_traffic.sh ${R}_cfg3_share cfg3 "score_quad_kernel reduce_buckets_kernel<COMPRESS>=reduce_buckets km_write_c_kernel=km_write_c" --config cfg3 --groups 125 > $OUT/${R}_cfg3_traffic.log 2>&1
bash tools/prof_traffic.sh ${R}_cfg4 cfg4 "score_xp_kernel<WRITE>=score_xp_kernel&true> score_xp_kernel<COUNT>=score_xp_kernel&false> reduce_ranges_kernel km_write_c_kernel km_count score_overflow_xp" --config cfg4 > $OUT/${R}_cfg4_traffic.log 2>&1
bash tools/prof_backhalf.sh ${R}_cfg3_share --config cfg3 --groups 125 > /dev/null 2>&1
bash tools/prof_backhalf.sh ${R}_cfg4 --config cfg4 > /dev/null 2>&1
fi
ls $OUT | grep "^${R}_" | tr '\n' ' '
