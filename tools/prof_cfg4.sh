#!/bin/bash
# usage: tools/prof_cfg4.sh <tag>  -- cfg4 (AA k=6): bench line, kernel stats, PMC traffic per kernel -> gpurun_out/<tag>_*
set -e
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 300 python3 bench.py --config cfg4 --steps 3 --warmup 1 --e2e 0 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_ks
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -- python3 $ROOT/bench.py --config cfg4 --steps 3 --warmup 1 --cpu-groups 0 --e2e 0 > $OUT/${TAG}_ks.log 2>&1
cp $(find $OUT/${TAG}_ks -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_ks
head -14 $OUT/${TAG}_kernel_stats.csv
bash $ROOT/tools/prof_traffic.sh ${TAG} cfg4 "score_xp_kernel<WRITE>=score_xp_kernel&true> score_xp_kernel<COUNT>=score_xp_kernel&false> reduce_ranges_kernel km_write_c_kernel km_count score_overflow_xp" --config cfg4
