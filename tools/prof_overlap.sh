#!/bin/bash
# usage: tools/prof_overlap.sh <tag>  -- kernel traces of tools/overlap_probe.py (halves in sequence / from two threads) -> gpurun_out/<tag>_overlap_timeline.json
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT && timeout -k 10 300 python3 tools/overlap_probe.py --steps 6 > $OUT/${TAG}_overlap_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_ovs $OUT/${TAG}_ovp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_ovs -- python3 $ROOT/tools/overlap_probe.py --steps 6 --phase seq > $OUT/${TAG}_ovs.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_ovp -- python3 $ROOT/tools/overlap_probe.py --steps 6 --phase par > $OUT/${TAG}_ovp.log 2>&1
python3 $ROOT/tools/overlap_timeline.py $OUT/${TAG}_overlap_timeline.json $OUT/${TAG}_ovs $OUT/${TAG}_ovp
cat $OUT/${TAG}_overlap_probe.txt
rm -rf $OUT/${TAG}_ovs $OUT/${TAG}_ovp
