#!/bin/bash
# usage: tools/prof_traffic.sh <tag> <workload> "<kernel substrings, space separated>" [bench args...]
# two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE -- they do not fit one pass) -> gpurun_out/<tag>_pmc_traffic.json
set -e
TAG=$1; WL=$2; SUBS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_tf $OUT/${TAG}_tw
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_tf -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-groups 0 --e2e 0 "$@" > $OUT/${TAG}_tf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_tw -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-groups 0 --e2e 0 "$@" > $OUT/${TAG}_tw.log 2>&1
python3 $ROOT/tools/pmc_traffic.py $(find $OUT/${TAG}_tf -name '*counter_collection.csv' | head -1) $(find $OUT/${TAG}_tw -name '*counter_collection.csv' | head -1) "$WL" $OUT/${TAG}_pmc_traffic.json $SUBS
rm -rf $OUT/${TAG}_tf $OUT/${TAG}_tw
