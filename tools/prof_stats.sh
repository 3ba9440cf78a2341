#!/bin/bash
# usage: tools/prof_stats.sh <tag> [bench args...]  -- rocprofv3 kernel stats of bench.py -> gpurun_out/<tag>_kernel_stats.csv
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_ks
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -- python3 $ROOT/bench.py --steps 10 --warmup 2 --cpu-groups 0 --e2e 0 "$@" > $OUT/${TAG}_ks.log 2>&1
cp $(find $OUT/${TAG}_ks -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_ks
