#!/bin/bash
# usage: tools/prof_sq.sh <tag> [bench args...]   -- two SQ counter passes + kernel stats of bench.py, summaries under gpurun_out/<tag>_*
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
B="SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS"
rm -rf $OUT/${TAG}_pa $OUT/${TAG}_pb $OUT/${TAG}_ks
rocprofv3 --pmc $A --output-format csv -d $OUT/${TAG}_pa -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-groups 0 --e2e 0 "$@" > $OUT/${TAG}_pa.log 2>&1
rocprofv3 --pmc $B --output-format csv -d $OUT/${TAG}_pb -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-groups 0 --e2e 0 "$@" > $OUT/${TAG}_pb.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -- python3 $ROOT/bench.py --steps 5 --warmup 2 --cpu-groups 0 --e2e 0 "$@" > $OUT/${TAG}_ks.log 2>&1
python3 $ROOT/tools/sq_counters.py $OUT/${TAG}_sq.json $(find $OUT/${TAG}_pa $OUT/${TAG}_pb -name '*counter_collection.csv')
cp $(find $OUT/${TAG}_ks -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
head -12 $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_pa $OUT/${TAG}_pb $OUT/${TAG}_ks
