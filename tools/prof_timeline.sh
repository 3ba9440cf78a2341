#!/bin/bash
# usage: tools/prof_timeline.sh <tag> <cfg> <groups>  -- kernel + copy timeline of one build step -> gpurun_out/<tag>_timeline.txt
set -e
TAG=$1; CFG=$2; NG=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_tl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/${TAG}_tl -- python3 $ROOT/tools/step_trace.py $CFG $NG 6 > $OUT/${TAG}_tl.log 2>&1
python3 $ROOT/tools/timeline.py $OUT/${TAG}_tl > $OUT/${TAG}_timeline.txt
rm -rf $OUT/${TAG}_tl
