#!/bin/bash
# usage: tools/prof_api.sh <tag> <cfg> <groups>  -- HIP API + kernel timeline of one build step -> gpurun_out/<tag>_api.txt
set -e
TAG=$1; CFG=$2; NG=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_api
rocprofv3 --kernel-trace --hip-trace --output-format csv -d $OUT/${TAG}_api -- python3 $ROOT/tools/step_trace.py $CFG $NG 6 > $OUT/${TAG}_api.log 2>&1
python3 $ROOT/tools/api_timeline.py $OUT/${TAG}_api 3 > $OUT/${TAG}_api.txt
rm -rf $OUT/${TAG}_api
